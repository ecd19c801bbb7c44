"""Counterpart of the reference's estimation.py: `demo_estimation`, `auc`, `convert_coordinates`
keep their names and signatures (estimation.py:12-117) so test.py:66-71 and
benchmark/multimodal_homog_benchmark_multiscale.py:70 can call them unchanged.

The solve itself -- cv2.findHomography(pos_a, pos_b, cv2.RANSAC, confidence=0.99999,
ransacReprojThreshold=3) at estimation.py:66-72 -- runs on the device (csrc/homography.hip):
the matches never leave the GPU before H is known, and many pairs are solved per launch
(`estimate_homographies`).  OpenCV, kornia, matplotlib are not needed.
"""
import json
import time

import numpy as np
import torch

from . import ops

RANSAC_THRESHOLD = 3.0   # ransacReprojThreshold (estimation.py:71)
RANSAC_ITERS = 2000      # OpenCV's default maxIters
RANSAC_CONFIDENCE = 0.99999  # estimation.py:70
CORNER_ERROR_CLAMP = 70.0


def auc(errors, thresholds):
    """Area under the recall-vs-error curve up to each threshold, divided by it (estimation.py:12-24)."""
    errors = np.sort(np.asarray(errors, dtype=np.float64))
    n = len(errors)
    recall = np.concatenate(([0.0], (np.arange(n) + 1) / n))
    errors = np.concatenate(([0.0], errors))
    out = []
    for t in thresholds:
        k = int(np.searchsorted(errors, t))
        e = np.concatenate((errors[:k], [t]))
        r = np.concatenate((recall[:k], [recall[k - 1]]))
        out.append(float(np.sum(np.diff(e) * (r[1:] + r[:-1]) * 0.5) / t))
    return out


def convert_coordinates(im_A_coords, im_A_to_im_B, wq, hq, wsup, hsup):
    """Normalised [-1,1] -> pixel coordinates, (w-1)(x+1)/2 (estimation.py:26-45).  numpy or torch."""
    stack = torch.stack if isinstance(im_A_coords, torch.Tensor) else np.stack
    a = stack(((wq - 1) * (im_A_coords[..., 0] + 1) / 2, (hq - 1) * (im_A_coords[..., 1] + 1) / 2), -1)
    b = stack(((wsup - 1) * (im_A_to_im_B[..., 0] + 1) / 2, (hsup - 1) * (im_A_to_im_B[..., 1] + 1) / 2), -1)
    return a, b


def estimate_homographies(good_matches, sizes, thresh=RANSAC_THRESHOLD, iters=RANSAC_ITERS, seed=0, confidence=RANSAC_CONFIDENCE):
    """Batched device-side replacement of estimation.py:61-77 (RANSAC with OpenCV's confidence-driven iteration bound).
    good_matches: (Bt,N,4) or (N,4) normalised warp rows on the GPU; sizes = (w1,h1,w2,h2).
    Returns H (Bt,3,3) float64 on the device; failures are diag(0,0,1) like the reference."""
    w1, h1, w2, h2 = sizes
    pts = ops.convert_matches(good_matches, w1, h1, w2, h2)
    H, _, _ = ops.find_homography(pts, thresh=thresh, iters=iters, seed=seed, confidence=confidence)
    return H


def corner_error(H_gt, H_pred, w, h, clamp=CORNER_ERROR_CLAMP):
    """Mean distance of the four warped image corners, clamped to 70 (estimation.py:79-92)."""
    corners = np.array([[0, 0, 1], [0, h - 1, 1], [w - 1, 0, 1], [w - 1, h - 1, 1]], np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        a = corners @ np.asarray(H_gt, np.float32).astype(np.float64).T  # the reference keeps H_s2t in float32
        b = corners @ np.asarray(H_pred, np.float64).T
        d = float(np.mean(np.linalg.norm(a[:, :2] / a[:, 2:] - b[:, :2] / b[:, 2:], axis=1)))
    return clamp if d > clamp else d


def demo_estimation(model, img1_path, img2_path, H_s2t_path, if_print=False):
    """match -> sample(5000) -> homography -> mean corner error; returns (ACE, runtime) (estimation.py:46-92)."""
    from PIL import Image

    im_1, im_2 = Image.open(img1_path), Image.open(img2_path)
    with open(H_s2t_path, "r") as f:
        H_s2t = np.array(json.load(f)["H"], dtype=np.float32)
    w1, h1 = im_1.size
    w2, h2 = im_2.size
    start = time.time()
    dense_matches, dense_certainty = model.match(im_1, im_2)
    good_matches, _ = model.sample(dense_matches, dense_certainty, 5000)
    H_pred = estimate_homographies(good_matches, (w1, h1, w2, h2))[0].cpu().numpy()  # the only device->host copy
    runtime = time.time() - start
    mean_dist = corner_error(H_s2t, H_pred, w1, h1)
    if if_print:
        print(f"ACE is {mean_dist}.")
    return mean_dist, runtime
