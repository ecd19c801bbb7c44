"""CPU oracle for the GFNet hot path -- TEST INFRASTRUCTURE, not product code.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The heavy loops live in oracle/gfnet_oracle.c / homography_oracle.c (gcc, OpenMP; built by
oracle/build.py in an fp32 and an fp64 variant); the small elementwise steps are numpy below.
Every function cites the reference file:line it restates (paths relative to KN-Zhang/GFNet).
Pinned against reference-generated goldens by tests/test_oracle_golden.py.
"""
import ctypes
import os

import numpy as np

from . import build as _build

_LIBS = {}
_c_long = ctypes.c_long
_c_int = ctypes.c_int
_vp = ctypes.c_void_p


def lib(variant="f32"):
    if variant not in _LIBS:
        path = _build.lib_path(variant)
        if not os.path.exists(path):
            _build.build()
        L = ctypes.CDLL(path)
        L.oracle_real_bytes.restype = _c_int
        L.oracle_max_threads.restype = _c_int
        _LIBS[variant] = L
    return _LIBS[variant]


def _real(variant):
    return np.float32 if variant == "f32" else np.float64


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(_vp) if a is not None else None


def set_threads(n):
    for v in ("f32", "f64"):
        lib(v).oracle_set_threads(_c_int(n))


def max_threads():
    return lib("f32").oracle_max_threads()


# ------------------------------------------------------------------------------------------
def avg_pool2(x):
    """F.avg_pool2d(x, 2, 2) -- utils/local_correlation.py:71."""
    x = _f32(x)
    B, C, H, W = x.shape
    out = np.empty((B, C, H // 2, W // 2), np.float32)
    lib("f32").oracle_avg_pool2(_p(x), _p(out), _c_int(B * C), _c_int(H), _c_int(W))
    return out


def local_correlation(featuremap_size, feature0, feature1, local_radius, num_grid, padding_mode="zeros", flow=None,
                      im_A_coords=None, sample_mode="bilinear", grid_based_correlation=False, num_level=1,
                      variant="f32", out=None, out_channel_offset=0):
    """utils/local_correlation.py:4-72 (same signature).  `out`, if given, is a (B,Ctot,G,G)
    array of the oracle's real type and the K channels are written at out_channel_offset."""
    assert padding_mode == "zeros" and sample_mode == "bilinear"
    B, c, h, w = featuremap_size
    f0 = _f32(feature0)
    f1 = _f32(feature1)
    G = int(num_grid)
    r = int(local_radius)
    K1 = (2 * r + 1) ** 2
    K = K1 * num_level
    fl = _f32(flow) if flow is not None else None
    rt = _real(variant)
    if out is None:
        out = np.empty((B, K, G, G), rt)
        out_channel_offset = 0
    assert out.dtype == rt and out.flags.c_contiguous
    out_bs = out.shape[1] * G * G
    L = lib(variant)
    for level in range(num_level):
        _, _, hh, ww = f1.shape
        base = out.ctypes.data + (out_channel_offset + level * K1) * G * G * out.itemsize
        L.oracle_local_correlation(_p(f0), _c_long(c * G * G), _p(f1), _p(fl), _vp(base), _c_long(out_bs), _c_int(B),
                                   _c_int(c), _c_int(G), _c_int(hh), _c_int(ww), _c_int(r),
                                   _c_int(1 if grid_based_correlation else 0), _c_int(h), _c_int(w))
        if level + 1 < num_level:
            f1 = avg_pool2(f1)
    return out


def local_correlation_grad_feature0(featuremap_size, grad_out, feature1, local_radius, num_grid, flow=None,
                                    grid_based_correlation=False, num_level=1, variant="f64"):
    """d(sum(grad_out * local_correlation(f0, ...))) / d f0.  The forward is linear in feature0 with a
    per-cell, per-channel coefficient S[b,c,i,j,k] (local_correlation.py:60; nothing else gets a gradient, :54), so the
    gradient is read off the pinned forward itself: run it with feature0 = one-hot channel c to obtain S / sqrt(C) and
    contract with grad_out over the K taps."""
    B, c, h, w = [int(v) for v in featuremap_size]
    G = int(num_grid)
    g = np.asarray(grad_out, np.float64)
    out = np.zeros((B, c, G, G), np.float64)
    for ch in range(c):
        e = np.zeros((B, c, G, G), np.float32)
        e[:, ch] = 1.0
        s = local_correlation(featuremap_size, e, feature1, local_radius, G, flow=flow, grid_based_correlation=grid_based_correlation,
                              num_level=num_level, variant=variant)
        out[:, ch] = (g * s).sum(axis=1)
    return out


def corr_volume(feat0, feat1, variant="f32"):
    """model/network.py:415-428 -> (B,H1,W1,H0,W0)."""
    f0, f1 = _f32(feat0), _f32(feat1)
    B, C, H0, W0 = f0.shape
    _, _, H1, W1 = f1.shape
    vol = np.empty((B, H1, W1, H0, W0), _real(variant))
    lib(variant).oracle_corr_softargmax(_p(f0), _p(f1), _p(vol), None, _c_int(B), _c_int(C), _c_int(H0), _c_int(W0),
                                        _c_int(H1), _c_int(W1))
    return vol


def corr_softargmax(feat0, feat1, variant="f32"):
    """pos_embed(corr_volume(f0,f1)) fused -- model/network.py:415-440 -> flow (B,2,H0,W0)."""
    f0, f1 = _f32(feat0), _f32(feat1)
    B, C, H0, W0 = f0.shape
    _, _, H1, W1 = f1.shape
    flow = np.empty((B, 2, H0, W0), _real(variant))
    lib(variant).oracle_corr_softargmax(_p(f0), _p(f1), None, _p(flow), _c_int(B), _c_int(C), _c_int(H0), _c_int(W0),
                                        _c_int(H1), _c_int(W1))
    return flow


def pos_embed(vol):
    """model/network.py:430-440 on an explicit volume (numpy, float64 softmax)."""
    B, H1, W1, H0, W0 = vol.shape
    v = vol.reshape(B, H1 * W1, H0 * W0).astype(np.float64)
    v = v - v.max(axis=1, keepdims=True)
    e = np.exp(v)
    P = e / e.sum(axis=1, keepdims=True)
    xs = (np.arange(W1) * 2 + 1) / W1 - 1
    ys = (np.arange(H1) * 2 + 1) / H1 - 1
    gx, gy = np.meshgrid(xs, ys, indexing="xy")
    grid = np.stack((gx.reshape(-1), gy.reshape(-1)), -1)
    return np.einsum("bji,jd->bdi", P, grid).reshape(B, 2, H0, W0)


def kde(x, std=0.1, half=True, down=None, variant="f32"):
    """utils/kde.py:4-13.  half=True rounds the inputs to fp16 first (the reference then also
    runs cdist in fp16, which this oracle does not mimic: see DESIGN.md)."""
    x = np.asarray(x)
    if half:
        x = x.astype(np.float16)
    xf = _f32(x)
    y = np.ascontiguousarray(xf[::down]) if down is not None else xf
    out = np.empty((xf.shape[0],), _real(variant))
    lib(variant).oracle_kde(_p(xf), _c_int(xf.shape[0]), _p(y), _c_int(y.shape[0]), _c_int(xf.shape[1]),
                            ctypes.c_double(std), _p(out))
    return out


def grid_sample(x, grid, variant="f32"):
    """F.grid_sample(x, grid, mode='bilinear', align_corners=False) -- model/network.py:537,547."""
    x, grid = _f32(x), _f32(grid)
    B, C, H, W = x.shape
    _, Ho, Wo, _ = grid.shape
    out = np.empty((B, C, Ho, Wo), _real(variant))
    lib(variant).oracle_grid_sample(_p(x), _p(grid), _p(out), _c_long(C * Ho * Wo), _c_int(B), _c_int(C), _c_int(H),
                                    _c_int(W), _c_int(Ho), _c_int(Wo))
    return out


def interpolate_bilinear(x, size, variant="f32"):
    """F.interpolate(x, size=size, mode='bilinear', align_corners=False) -- model/network.py:238-249,271-281."""
    x = _f32(x)
    B, C, H, W = x.shape
    Ho, Wo = (size, size) if np.isscalar(size) else size
    out = np.empty((B, C, Ho, Wo), _real(variant))
    lib(variant).oracle_interp_bilinear(_p(x), _p(out), _c_int(B * C), _c_int(H), _c_int(W), _c_int(Ho), _c_int(Wo))
    return out


def interpolate_bicubic(x, size, variant="f32"):
    """F.interpolate(x, size=size, mode='bicubic', align_corners=False, antialias=False)."""
    x = _f32(x)
    B, C, H, W = x.shape
    Ho, Wo = (size, size) if np.isscalar(size) else size
    out = np.empty((B, C, Ho, Wo), _real(variant))
    lib(variant).oracle_interp_bicubic(_p(x), _p(out), _c_int(B * C), _c_int(H), _c_int(W), _c_int(Ho), _c_int(Wo))
    return out


IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def resize_normalise(im, size, mode="bicubic", variant="f32"):
    """get_tuple_transform_ops(resize=size, mode, normalize=True) on a (B,3+,H,W) float image in [0,1] --
    utils/utils.py:18-27, 87-116 as GFNet.match calls it (model/network.py:293-346): torchvision Resize on a tensor with
    antialias=None = F.interpolate(mode, align_corners=False, antialias=False); mode 'bilinear' for path inputs and the
    upsample pass (the reference passes mode=2), 'bicubic' (the default) for PIL / tensor inputs; then (x - mean) / std
    on the first three channels."""
    x = _f32(im)[:, :3]
    f = {"bicubic": interpolate_bicubic, "bilinear": interpolate_bilinear}[mode]
    y = f(x, size, variant) if tuple(x.shape[-2:]) != tuple(size) else x.astype(_real(variant))
    rt = _real(variant)
    mean = np.asarray(IMAGENET_MEAN, rt).reshape(1, 3, 1, 1)
    std = np.asarray(IMAGENET_STD, rt).reshape(1, 3, 1, 1)
    return ((y - mean) / std).astype(rt)


def _linspace_f32(start, end, steps):
    """torch.linspace in float32 (ATen: symmetric fill from both ends)."""
    start, end = np.float32(start), np.float32(end)
    if steps == 1:
        return np.array([start], np.float32)
    step = np.float32((end - start) / np.float32(steps - 1))
    i = np.arange(steps)
    lo = start + step * i.astype(np.float32)
    hi = end - step * (steps - 1 - i).astype(np.float32)
    return np.where(i < steps // 2, lo, hi).astype(np.float32)


def cell_centres(G):
    """linspace(-1+1/G, 1-1/G, G) -- the A-image grid coordinates (model/network.py:539-546, 362-367)."""
    return _linspace_f32(-1 + 1 / G, 1 - 1 / G, G)


def refiner_input(num_grid, x, y, flow, disp_w, disp_b, local_radius, scale_factor=1.0, corr_in_other=True,
                  variant="f32"):
    """ConvRefiner.forward up to the concat -- model/network.py:533-558.
    Returns d = cat(grid_feature, x_hat, disp_emb(40/32*scale_factor*(flow-grid)), local_corr)."""
    x, y, flow = _f32(x), _f32(y), _f32(flow)
    b, c, hs, ws = x.shape
    G = int(num_grid)
    rt = _real(variant)
    x_hat = grid_sample(y, np.ascontiguousarray(flow.transpose(0, 2, 3, 1)), variant)  # :537
    lin = cell_centres(G)
    gy, gx = np.meshgrid(lin, lin, indexing="ij")
    coords = np.broadcast_to(np.stack((gx, gy))[None], (b, 2, G, G)).astype(np.float32)  # :539-546
    grid_feature = grid_sample(x, np.ascontiguousarray(coords.transpose(0, 2, 3, 1)), variant)  # :547
    in_disp = (np.float32(40 / 32 * scale_factor) * (flow - coords)).astype(rt)  # :548-549
    w = np.asarray(disp_w, rt).reshape(-1, 2)
    emb = np.einsum("od,bdij->boij", w, in_disp) + np.asarray(disp_b, rt)[None, :, None, None]
    parts = [grid_feature, x_hat, emb.astype(rt)]
    if corr_in_other:
        lc = local_correlation((b, c, hs, ws), grid_feature.astype(np.float32), y, local_radius, G, flow=flow,
                               variant=variant)  # :553-554
        parts.append(lc)
    return np.concatenate(parts, axis=1)  # :555 / :558


def flow_update(flow, certainty, delta_flow, delta_cert, disp_prev, scale, W0, H0, training=False, return_rel=False):
    """model/network.py:262-268: displacement scaling, eval-time zeroing, accumulation.
    Returns (flow, certainty, displacement) [, rel]: rel = |d - d_prev| / |d_prev|, the quantity the eval-time rule compares
    with 1e-6 (tests use it to find the cells where that discontinuous rule is decided by the last bit)."""
    f32 = np.float32
    # reference: int(scale) * stack(dx/(4*W0), dy/(4*H0))
    d = np.stack((delta_flow[:, 0].astype(f32) / f32(4 * W0), delta_flow[:, 1].astype(f32) / f32(4 * H0)), axis=1)
    d = (f32(int(scale)) * d).astype(f32)
    if not training:
        with np.errstate(divide="ignore", invalid="ignore"):
            rel = np.abs(d - disp_prev) / np.abs(disp_prev)
        d = np.where(rel < f32(1e-6), f32(0), d)
    else:
        rel = np.full_like(d, np.inf)
    out = ((flow + d).astype(f32), (certainty + delta_cert).astype(f32), d)
    return out + (rel,) if return_rel else out


def match_post(flow, certainty, low_res_certainty16=None, symmetric=True, attenuate_cert=True):
    """model/network.py:332-338 + 358-384.  flow (nb,2,G,G), certainty (nb,1,G,G) finest-scale
    logits; low_res_certainty16: the scale-16 certainty (nb,1,Gc,Gc) or None.
    Returns warp (B,G,2G,4) / (B,G,G,4) and certainty (B,G,2G) / (B,G,G) (batched form)."""
    f32 = np.float32
    nb, _, G, _ = flow.shape
    cert = certainty.astype(f32)
    if attenuate_cert:
        low = interpolate_bilinear(low_res_certainty16, (G, G)).astype(f32)
        low = f32(0.5) * low * (low < 0)
        cert = cert - low
    cert = (1.0 / (1.0 + np.exp(-cert.astype(np.float64)))).astype(f32)
    fl = flow.transpose(0, 2, 3, 1)
    wrong = (np.abs(fl) > 1).sum(-1) > 0
    cert = np.where(wrong[:, None], f32(0), cert)
    fl = np.clip(fl, -1, 1)
    lin = cell_centres(G)
    gx, gy = np.meshgrid(lin, lin, indexing="xy")
    B = nb // 2 if symmetric else nb
    grid = np.broadcast_to(np.stack((gx, gy), -1)[None], (B, G, G, 2))
    if symmetric:
        a2b, b2a = fl[:B], fl[B:]
        q = np.concatenate((grid, a2b), -1)
        s = np.concatenate((b2a, grid), -1)
        warp = np.concatenate((q, s), 2)
        cert = np.concatenate((cert[:B], cert[B:]), 3)
    else:
        warp = np.concatenate((grid, fl), -1)
    return warp.astype(f32), cert[:, 0]


def sample(matches, certainty, num=5000, sample_mode="threshold_balanced", sample_thresh=0.05, device_is_gpu=False):
    """model/network.py:385-414 on the CPU with torch's CPU generator (torch.multinomial is the
    reference's own RNG consumer; seeding torch reproduces the reference's draws)."""
    import torch

    m = torch.as_tensor(np.asarray(matches, np.float32)).reshape(-1, 4)
    c = torch.as_tensor(np.asarray(certainty, np.float32)).clone().reshape(-1)
    if "threshold" in sample_mode:
        c[c > sample_thresh] = 1
    expansion = 4 if "balanced" in sample_mode else 1
    good = torch.multinomial(c, num_samples=min(expansion * num, len(c)), replacement=False)
    gm, gc = m[good], c[good]
    if "balanced" not in sample_mode:
        return gm.numpy(), gc.numpy()
    density = torch.from_numpy(kde(gm.numpy(), std=0.1, half=device_is_gpu, down=(1 if device_is_gpu else 8)))
    p = 1 / (density + 1)
    p[density < 10] = 1e-7
    bal = torch.multinomial(p, num_samples=min(num, len(gc)), replacement=False)
    return gm[bal].numpy(), gc[bal].numpy()


def convert_coordinates(a, b, wq, hq, wsup, hsup):
    """estimation.py:26-45."""
    pa = np.stack(((wq - 1) * (a[..., 0] + 1) / 2, (hq - 1) * (a[..., 1] + 1) / 2), axis=-1)
    pb = np.stack(((wsup - 1) * (b[..., 0] + 1) / 2, (hsup - 1) * (b[..., 1] + 1) / 2), axis=-1)
    return pa, pb


def corner_error(H_gt, H_pred, w, h, clamp=70.0):
    """estimation.py:79-92 (ACE: mean corner error, clamped to 70).  The reference holds the
    ground truth as a float32 tensor (estimation.py:51), so H_gt is rounded to fp32 first."""
    H_gt = np.asarray(H_gt, np.float32)
    corners = np.array([[0, 0, 1], [0, h - 1, 1], [w - 1, 0, 1], [w - 1, h - 1, 1]], np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        a = corners @ np.asarray(H_gt, np.float64).T
        a = a[:, :2] / a[:, 2:]
        b = corners @ np.asarray(H_pred, np.float64).T
        b = b[:, :2] / b[:, 2:]
        d = float(np.mean(np.linalg.norm(a - b, axis=1)))
    if d > clamp or not np.isfinite(d):
        # nan > 70 is False in the reference (nan stays nan); keep that for non-finite H
        return clamp if d > clamp else d
    return d


def auc(errors, thresholds):
    """estimation.py:12-24."""
    errors = np.sort(np.asarray(errors, np.float64))
    recall = (np.arange(len(errors)) + 1) / len(errors)
    errors = np.r_[0.0, errors]
    recall = np.r_[0.0, recall]
    out = []
    for t in thresholds:
        last = np.searchsorted(errors, t)
        r = np.r_[recall[:last], recall[last - 1]]
        e = np.r_[errors[:last], t]
        out.append(float(np.sum((e[1:] - e[:-1]) * (r[1:] + r[:-1]) / 2) / t))
    return out


# ---- homography solve (estimation.py:60-77; OpenCV pipeline restated -- parity with OpenCV unpinned) ----
def convert_matches(matches, wA, hA, wB, hB):
    """estimation.py:26-45 in float32 (numpy's own evaluation order): (N,4) normalised -> (N,4) pixels."""
    m = np.ascontiguousarray(matches, np.float32).reshape(-1, 4)
    pts = np.empty_like(m)
    lib("f64").oracle_convert_matches(_p(m), _p(pts), _c_long(m.shape[0]), ctypes.c_float(wA), ctypes.c_float(hA),
                                      ctypes.c_float(wB), ctypes.c_float(hB))
    return pts.reshape(np.shape(matches))


def homography_dlt(pts, weight=None):
    """Weighted normalised DLT ("grid-DLT"): pts (Bt,N,4) pixel (x,y,u,v) -> H (Bt,3,3), ok (Bt,)."""
    pts = np.ascontiguousarray(pts, np.float32)
    Bt, N, _ = pts.shape
    w = None if weight is None else np.ascontiguousarray(weight, np.float64)
    H = np.empty((Bt, 9), np.float64)
    ok = np.empty((Bt,), np.int32)
    lib("f64").oracle_homography_dlt(_p(pts), _p(w), _c_int(Bt), _c_int(N), _p(H), _p(ok))
    return H.reshape(Bt, 3, 3), ok


def homography_ransac(pts, thresh=3.0, iters=2000, seed=0, lm_iters=10, stage=0, return_mask=False, confidence=0.99999,
                      return_iters=False):
    """RANSAC(4-pt) -> DLT on inliers -> LM (the findHomography pipeline, estimation.py:66-72: confidence=0.99999 shrinks
    the iteration bound with the best inlier ratio; confidence=0 scores all `iters` hypotheses).  pts (Bt,N,4) pixels.
    Returns H (Bt,3,3), inlier counts (Bt,), chosen hypothesis (Bt,) [, mask (Bt,N)] [, iteration bound at exit (Bt,)]."""
    pts = np.ascontiguousarray(pts, np.float32)
    Bt, N, _ = pts.shape
    H = np.empty((Bt, 9), np.float64)
    ninl = np.empty((Bt,), np.int32)
    best = np.empty((Bt,), np.int32)
    used = np.empty((Bt,), np.int32)
    mask = np.empty((Bt, N), np.uint8) if return_mask else None
    lib("f64").oracle_homography_ransac(_p(pts), _c_int(Bt), _c_int(N), ctypes.c_double(thresh), _c_int(iters),
                                        ctypes.c_double(confidence or 0.0), ctypes.c_uint64(seed), _c_int(lm_iters), _c_int(stage),
                                        _p(H), _p(ninl), _p(best), _p(mask), _p(used))
    out = (H.reshape(Bt, 3, 3), ninl, best)
    if return_mask:
        out = out + (mask,)
    return out + (used,) if return_iters else out


def homography_dlt_svd(pts, weight=None):
    """Independent float64 numpy-SVD DLT with Hartley normalisation (cross-check for the tests)."""
    pts = np.asarray(pts, np.float64)
    w = np.ones(len(pts)) if weight is None else np.asarray(weight, np.float64)

    def norm(p):
        c = (w[:, None] * p).sum(0) / w.sum()
        s = np.sqrt(2) / ((w * np.linalg.norm(p - c, axis=1)).sum() / w.sum())
        T = np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1]])
        return (p - c) * s, T

    a, Ta = norm(pts[:, :2])
    b, Tb = norm(pts[:, 2:])
    sw = np.sqrt(w)
    rows = []
    for (x, y), (u, v), s in zip(a, b, sw):
        rows.append(s * np.array([x, y, 1, 0, 0, 0, -u * x, -u * y, -u]))
        rows.append(s * np.array([0, 0, 0, x, y, 1, -v * x, -v * y, -v]))
    _, _, Vt = np.linalg.svd(np.array(rows))
    H = np.linalg.inv(Tb) @ Vt[-1].reshape(3, 3) @ Ta
    return H / H[2, 2]


# ---- refiner conv stack ---------------------------------------------------------
def conv_block(x, dw_w, dw_b, bn_weight, bn_bias, bn_mean, bn_var, pw_w, pw_b, eps=1e-5, variant="f32"):
    """One ConvRefiner block in eval mode -- model/network.py:471-487 (create_block):
    Conv2d(C, C, 5, padding 2, groups=C) -> BatchNorm2d (running statistics) -> ReLU -> Conv2d(C, M, 1).
    x (B,C,G,G); dw_w (C,1,5,5) or (C,25); dw_b (C) or None; pw_w (M,C[,1,1]); pw_b (M)."""
    rt = _real(variant)
    x = np.asarray(x, rt)
    B, C, G, G2 = x.shape
    w = np.asarray(dw_w, rt).reshape(C, 5, 5)
    xp = np.zeros((B, C, G + 4, G2 + 4), rt)
    xp[:, :, 2:-2, 2:-2] = x
    t = np.zeros_like(x)
    for dy in range(5):          # cross-correlation, as torch's conv2d: out[i,j] += w[dy,dx] * x[i+dy-2, j+dx-2]
        for dx in range(5):
            t += w[None, :, dy, dx, None, None] * xp[:, :, dy:dy + G, dx:dx + G2]
    if dw_b is not None:
        t += np.asarray(dw_b, rt)[None, :, None, None]
    inv = (1.0 / np.sqrt(np.asarray(bn_var, rt) + rt(eps))).astype(rt)       # ATen batch_norm, eval
    t = (t - np.asarray(bn_mean, rt)[None, :, None, None]) * inv[None, :, None, None] * np.asarray(bn_weight, rt)[None, :, None, None] \
        + np.asarray(bn_bias, rt)[None, :, None, None]
    t = np.maximum(t, 0)
    M = np.asarray(pw_w).shape[0]
    y = np.einsum("mc,bcij->bmij", np.asarray(pw_w, rt).reshape(M, C), t) + np.asarray(pw_b, rt)[None, :, None, None]
    return y.astype(rt)


def conv_stack(d, sd, variant="f32"):
    """out_conv(hidden_blocks(block1(d))) -- ConvRefiner.forward, model/network.py:560-563, from a
    ConvRefiner state_dict `sd` (reference parameter names: block1.{0,1,3}.*, hidden_blocks.N.{0,1,3}.*,
    out_conv.*).  Returns (B, out_dim, G, G)."""
    def block(prefix, h):
        return conv_block(h, sd[prefix + ".0.weight"], sd.get(prefix + ".0.bias"), sd[prefix + ".1.weight"], sd[prefix + ".1.bias"],
                          sd[prefix + ".1.running_mean"], sd[prefix + ".1.running_var"], sd[prefix + ".3.weight"],
                          sd[prefix + ".3.bias"], variant=variant)
    h = block("block1", d)
    n = 0
    while f"hidden_blocks.{n}.0.weight" in sd:
        h = block(f"hidden_blocks.{n}", h)
        n += 1
    rt = _real(variant)
    ow = np.asarray(sd["out_conv.weight"], rt)
    ow = ow.reshape(ow.shape[0], -1)
    return (np.einsum("mc,bcij->bmij", ow, h) + np.asarray(sd["out_conv.bias"], rt)[None, :, None, None]).astype(rt)
