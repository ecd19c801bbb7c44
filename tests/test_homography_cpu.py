"""Homography-solve oracle on known-H synthetic correspondences.

The reference's solver is cv2.findHomography (absent; parity with OpenCV unpinned -- see
oracle/homography_oracle.c).  These tests pin the restated pipeline by ground truth instead:
noise-free correspondences must give the true H to < 1e-3 px corner error, the weighted DLT must
agree with an independent numpy-SVD DLT, RANSAC must survive 40 % outliers."""
import numpy as np
import pytest

import oracle

S = 448


def random_h(rng, size=S, amp=0.15):
    """4-corner perturbation homography (same family as datasets/generate_random_H_large_size.py:6-36)."""
    src = np.array([[0, 0], [size - 1, 0], [size - 1, size - 1], [0, size - 1]], np.float64)
    dst = src + rng.uniform(-amp * size, amp * size, size=(4, 2))
    A = []
    for (x, y), (u, v) in zip(src, dst):
        A.append([x, y, 1, 0, 0, 0, -u * x, -u * y, -u])
        A.append([0, 0, 0, x, y, 1, -v * x, -v * y, -v])
    _, _, Vt = np.linalg.svd(np.array(A))
    H = Vt[-1].reshape(3, 3)
    return H / H[2, 2]


def make_points(rng, H, n, noise=0.0, outliers=0.0, size=S):
    a = rng.uniform(0, size - 1, size=(n, 2))
    ah = np.c_[a, np.ones(n)] @ H.T
    b = ah[:, :2] / ah[:, 2:]
    b = b + noise * rng.standard_normal((n, 2))
    k = int(outliers * n)
    if k:
        b[:k] = rng.uniform(0, size - 1, size=(k, 2))
    return np.c_[a, b].astype(np.float32)


def ace(Ha, Hb):
    return oracle.corner_error(Ha, Hb, S, S, clamp=1e9)


def test_dlt_noise_free_recovers_h():
    rng = np.random.default_rng(1)
    Hs = [random_h(rng) for _ in range(4)]
    pts = np.stack([make_points(rng, H, 500) for H in Hs])
    Hest, ok = oracle.homography_dlt(pts)
    assert ok.all()
    for H, He in zip(Hs, Hest):
        # float32 input coordinates limit this to ~1e-4 px
        assert ace(H.astype(np.float32), He) < 1e-3


def test_dlt_matches_numpy_svd_on_noisy_weighted_points():
    rng = np.random.default_rng(2)
    H = random_h(rng)
    pts = make_points(rng, H, 800, noise=0.7)
    w = rng.uniform(0.1, 1.0, size=800)
    He, ok = oracle.homography_dlt(pts[None], w[None])
    assert ok[0]
    # the oracle follows OpenCV's per-axis mean-abs-deviation normalisation, the cross-check uses
    # Hartley's isotropic one: same algebraic minimiser up to the normalisation -> close, not equal
    assert ace(oracle.homography_dlt_svd(pts, w), He[0]) < 0.05
    # unweighted, noise-free: both are exact
    pts0 = make_points(rng, H, 300)
    He0, _ = oracle.homography_dlt(pts0[None])
    assert ace(oracle.homography_dlt_svd(pts0), He0[0]) < 1e-3


def test_ransac_with_outliers_and_noise():
    rng = np.random.default_rng(3)
    Hs = [random_h(rng) for _ in range(3)]
    pts = np.stack([make_points(rng, H, 2000, noise=0.5, outliers=0.4) for H in Hs])
    Hest, ninl, best, mask = oracle.homography_ransac(pts, thresh=3.0, iters=500, seed=7, return_mask=True)
    for b, H in enumerate(Hs):
        assert best[b] >= 0 and ninl[b] > 1100
        assert mask[b].sum() == ninl[b]
        assert ace(H, Hest[b]) < 0.25  # 0.5 px noise on 1200 inliers
    # stages: LM must not be worse than the inlier DLT in reprojection error
    Hd, _, _ = oracle.homography_ransac(pts, thresh=3.0, iters=500, seed=7, stage=2)

    def rms(H, p, m):
        ph = np.c_[p[:, :2], np.ones(len(p))] @ H.T
        return np.sqrt((((ph[:, :2] / ph[:, 2:]) - p[:, 2:]) ** 2).sum(1)[m.astype(bool)].mean())

    for b in range(3):
        assert rms(Hest[b], pts[b], mask[b]) <= rms(Hd[b], pts[b], mask[b]) + 1e-9


def test_ransac_is_deterministic_in_seed_and_noise_free_exact():
    rng = np.random.default_rng(4)
    H = random_h(rng)
    pts = make_points(rng, H, 1000)[None]
    H1, n1, b1 = oracle.homography_ransac(pts, iters=64, seed=11)
    H2, n2, b2 = oracle.homography_ransac(pts, iters=64, seed=11)
    np.testing.assert_array_equal(H1, H2)
    assert n1[0] == 1000 and b1[0] == b2[0] == 0  # every hypothesis is perfect: the first one wins ties
    assert ace(H.astype(np.float32), H1[0]) < 1e-3
    H3, _, b3 = oracle.homography_ransac(pts, iters=64, seed=12)
    assert b3[0] == 0


def test_failure_convention_diag001():
    # fewer than 4 points / all-degenerate input -> H = diag(0,0,1) (estimation.py:74-76)
    pts = np.zeros((1, 3, 4), np.float32)
    H, n, b = oracle.homography_ransac(pts, iters=16)
    np.testing.assert_array_equal(H[0], np.diag([0.0, 0.0, 1.0]))
    assert b[0] == -1
    pts = np.ones((1, 50, 4), np.float32)  # all points identical: every 4-point system is singular
    H, n, b = oracle.homography_ransac(pts, iters=16)
    np.testing.assert_array_equal(H[0], np.diag([0.0, 0.0, 1.0]))


def test_convert_matches_is_float32_numpy_formula():
    rng = np.random.default_rng(5)
    m = rng.uniform(-1, 1, size=(100, 4)).astype(np.float32)
    pts = oracle.convert_matches(m, 640, 480, 320, 200)
    pa, pb = oracle.convert_coordinates(m[:, :2], m[:, 2:], 640, 480, 320, 200)
    np.testing.assert_array_equal(pts[:, :2], pa.astype(np.float32))
    np.testing.assert_array_equal(pts[:, 2:], pb.astype(np.float32))


def test_iteration_bound_follows_opencv_rule():
    """cv::RANSACUpdateNumIters(confidence = 0.99999, modelPoints = 4): the loop stops at log(1e-5) / log(1 - w^4) once a
    hypothesis with inlier ratio w has been seen (estimation.py:66-72 passes that confidence)."""
    rng = np.random.default_rng(3)
    H = random_h(rng, 448)
    for outl, expect in ((0.0, 1), (0.3, 42), (0.6, 444)):
        pts = make_points(rng, H, 4000, noise=0.0, outliers=outl)[None]
        _, n, b, used = oracle.homography_ransac(pts, iters=2000, seed=9, return_iters=True)
        w = n[0] / 4000.0
        bound = np.log(1e-5) / np.log(1 - w ** 4) if w < 1 else 0
        assert abs(int(used[0]) - max(int(round(bound)), 0)) <= 1 or used[0] == 2000, (outl, used, bound)
        assert abs(used[0] - expect) <= 0.35 * expect + 2, (outl, used)
        assert 0 <= b[0] < max(used[0], 1)
    # confidence 0: every hypothesis is scored
    _, _, _, used = oracle.homography_ransac(pts, iters=300, seed=9, confidence=0, return_iters=True)
    assert used[0] == 300
