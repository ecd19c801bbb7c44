"""Thin Python bindings of the C-ABI entry points (include/gfnet_hip.h) on torch device tensors.

Each function cites the reference code it stands in for (paths relative to KN-Zhang/GFNet).
torch supplies device memory and the current HIP stream; every result comes from csrc/*.hip.
There is no CPU implementation here: CPU tensors raise.
"""
import math

import torch

from . import _lib
from ._lib import c_vp, check, f32c, featc, ptr, require_gpu, stream_ptr


def _L():
    return _lib.lib()


# bench.py sets this to a dict {name: [(start_event, end_event), ...]} to time individual launches
# with HIP events on the launch stream; None (the default) costs nothing.
kernel_events = None
# bench.py / tools set this to a dict to collect, per local-correlation call (name -> [(tiles left to the second launch,
# cells redone per tap, tiles staged in halves), ...]), the counters the kernels leave in the scratch header; costs a device sync per call.
kernel_counters = None
# the kernels read fp16 feature maps directly (BASELINE config 5): no widened copy is made
NATIVE_FP16 = True
# Large windows (r >= 5 on 64-channel maps) multiply on the matrix core with split-bf16 operands by default: every product is exact to
# 2^-17 relative, so a correlation value is within 2^-17 * sum_c |f0_c * f1_c| / sqrt(C) of the fp32 result (a few 1e-6 on unit-scale
# features, but proportional to the operands' magnitude, not to the result's: strongly cancelling sums lose relative accuracy).
# True keeps every radius on the fp32 FMA kernels (C-ABI variant 4), bit-identical to the round-1 kernel.
LOCAL_CORR_FP32 = False
# refiner_input writes the tile plan of the local correlation that follows it from extra workgroups of its own launch.  bench.py
# switches this off for a few untimed steps so that the plan becomes the correlation call's own first launch and lands inside its
# event bracket (`roofline.frac_incl_plan`).
FUSE_PLAN = True


def _timed(name, launch):
    if kernel_events is None or name not in kernel_events:
        return launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = launch()
    e1.record()
    kernel_events[name].append((e0, e1))
    return r


def corr_softargmax(feat0, feat1, symmetric=False):
    """pos_embed(corr_volume(feat0, feat1)) without writing the volume (model/network.py:251-252, 415-440).
    feat0 (B,C,H0,W0), feat1 (B,C,H1,W1) -> flow (B,2,H0,W0).  symmetric=True: the result has 2B
    directions, (feat0 vs feat1) then (feat1 vs feat0) -- the reference's concatenated batch
    (network.py:213-222) without copying the features."""
    dev = require_gpu(feat0, feat1)
    (f0, dt0), (f1, dt1) = featc(feat0), featc(feat1)
    if dt0 != dt1:
        f0, f1, dt0 = f32c(f0), f32c(f1), _lib.GFN_F32
    B, C, H0, W0 = f0.shape
    B1, C1, H1, W1 = f1.shape
    if B1 != B or C1 != C:
        raise ValueError("feat0/feat1 batch or channel mismatch")
    nb = 2 * B if symmetric else B
    flow = torch.empty((nb, 2, H0, W0), device=dev, dtype=torch.float32)
    # workspace of the split-bf16 path (64-channel maps): the caller owns every buffer (include/gfnet_hip.h); a fresh tensor per call
    # (the caching allocator hands the same block back; not the per-stream scratch, whose header the local correlation keeps zeroed)
    nws = int(_L().gfn_corr_softargmax_ws_bytes(nb, C, H1, W1))
    ws = torch.empty(nws, device=dev, dtype=torch.uint8) if nws > 0 else None
    check(_L().gfn_corr_softargmax_fwd_ws(ptr(f0), ptr(f1), dt0, ptr(flow), nb, C, H0, W0, H1, W1, 1 if symmetric else 0,
                                          ptr(ws), nws, stream_ptr(dev)), "gfn_corr_softargmax_fwd")
    return flow


def corr_volume(feat0, feat1, with_flow=False):
    """GFNet.corr_volume (model/network.py:415-428): (B,H1,W1,H0,W0) = f0^T f1 / sqrt(C)."""
    dev = require_gpu(feat0, feat1)
    f0, f1 = f32c(feat0), f32c(feat1)
    B, C, H0, W0 = f0.shape
    _, _, H1, W1 = f1.shape
    vol = torch.empty((B, H1, W1, H0, W0), device=dev, dtype=torch.float32)
    flow = torch.empty((B, 2, H0, W0), device=dev, dtype=torch.float32) if with_flow else None
    check(_L().gfn_corr_volume_fwd(ptr(f0), ptr(f1), ptr(vol), ptr(flow), B, C, H0, W0, H1, W1, stream_ptr(dev)),
          "gfn_corr_volume_fwd")
    return (vol, flow) if with_flow else vol


def pos_embed(corr_vol):
    """GFNet.pos_embed (model/network.py:430-440) on an explicit volume (B,H1,W1,H0,W0) -> (B,2,H0,W0)."""
    dev = require_gpu(corr_vol)
    v = f32c(corr_vol)
    B, H1, W1, H0, W0 = v.shape
    flow = torch.empty((B, 2, H0, W0), device=dev, dtype=torch.float32)
    check(_L().gfn_pos_embed_fwd(ptr(v), ptr(flow), B, H0, W0, H1, W1, stream_ptr(dev)), "gfn_pos_embed_fwd")
    return flow


def refiner_input(num_grid, x, y, flow, disp_w, disp_b, local_radius, scale_factor=1.0, corr_in_other=True, reuse=None):
    """The concat tensor `d` of ConvRefiner.forward (model/network.py:533-558):
    cat(grid_sample(x, cell centres), grid_sample(y, flow), disp_emb(40/32*scale_factor*(flow-centres)),
    local_correlation(...)) -- every slice written in place by the HIP kernels, no torch.cat.
    If flow has twice the batch of x/y the call is symmetric: directions (x vs y) then (y vs x).
    reuse: the `d` an earlier call returned for the SAME x and num_grid (the previous refiner iteration at this scale): it is
    overwritten in place except for its grid_feature planes, which depend on x and the grid only."""
    dev = require_gpu(x, y, flow, disp_w, disp_b)
    (x, dtx), (y, dty), fl = featc(x), featc(y), f32c(flow)
    if dtx != dty:
        x, y, dtx = f32c(x), f32c(y), _lib.GFN_F32
    Bi, C, Hs, Ws = x.shape
    G = int(num_grid)
    B = fl.shape[0]
    symmetric = B == 2 * Bi
    if tuple(y.shape) != (Bi, C, Hs, Ws) or tuple(fl.shape[1:]) != (2, G, G) or B not in (Bi, 2 * Bi):
        raise ValueError(f"refiner_input: y must be {(Bi, C, Hs, Ws)} and flow (B or 2B,2,{G},{G}), got {tuple(y.shape)}, {tuple(fl.shape)}")
    w = f32c(disp_w).reshape(-1, 2)
    bvec = f32c(disp_b).reshape(-1)
    Dd = w.shape[0]
    r = int(local_radius)
    K = (2 * r + 1) ** 2 if corr_in_other else 0
    CH = 2 * C + Dd + K
    keep = reuse is not None
    if keep:
        if tuple(reuse.shape) != (B, CH, G, G) or reuse.dtype != torch.float32 or not reuse.is_contiguous() or reuse.device != dev:
            raise ValueError("refiner_input: `reuse` must be the tensor an earlier call with the same shapes returned")
        d = reuse
    else:
        d = torch.empty((B, CH, G, G), device=dev, dtype=torch.float32)
    st = stream_ptr(dev)
    mode = (1 if symmetric else 0) | (2 if keep else 0)  # include/gfnet_hip.h: GFN_RI_KEEP_GRID_FEATURE
    disp_scale = float(40 / 32 * scale_factor)
    # shapes the lean local-correlation path takes are planned inside the refiner-input launch (both only read the flow)
    plans = FUSE_PLAN and corr_in_other and bool(_L().gfn_local_corr_plans(C, Hs, Ws, G, r, dtx))
    if corr_in_other:
        nscr = int(_L().gfn_local_corr_scratch_bytes(B, G))
        scr = _lib.scratch(dev, nscr)
    if plans:
        check(_L().gfn_refiner_input_plan_fwd_dt(ptr(x), ptr(y), dtx, ptr(fl), ptr(w), ptr(bvec), ptr(d), CH * G * G, B, C, Hs, Ws, G, Dd,
                                                 disp_scale, mode, r, ptr(scr), nscr, st), "gfn_refiner_input_plan_fwd")
    else:
        check(_L().gfn_refiner_input_fwd_dt(ptr(x), ptr(y), dtx, ptr(fl), ptr(w), ptr(bvec), ptr(d), CH * G * G, B, C, Hs, Ws, G, Dd,
                                            disp_scale, mode, st), "gfn_refiner_input_fwd")
    if corr_in_other:
        out = d[:, 2 * C + Dd:]
        name = f"local_corr_c{C}_h{Hs}_g{G}_r{r}"
        check(_timed(name, lambda: _L().gfn_local_corr_fwd_dt(ptr(d), CH * G * G, ptr(y), ptr(x) if symmetric else None, dtx, ptr(fl),
                                                              c_vp(out.data_ptr()), CH * G * G, B, C, G, Hs, Ws, r, 0, Hs, Ws,
                                                              (8 if plans else 0) | (4 if LOCAL_CORR_FP32 and r >= 5 else 0),
                                                              ptr(scr), nscr, st)), "gfn_local_corr_fwd")
        if kernel_counters is not None:
            hdr = scr[:8].cpu()  # synchronises; header layout: csrc/local_corr.hip kTodoHdr
            kernel_counters.setdefault(name, []).append((int(hdr[3]), int(hdr[5]), int(hdr[7])))
    return d


def grid_sample(x, grid):
    """F.grid_sample(x, grid, mode='bilinear', padding_mode='zeros', align_corners=False)."""
    dev = require_gpu(x, grid)
    x, g = f32c(x), f32c(grid)
    B, C, H, W = x.shape
    _, Ho, Wo, _ = g.shape
    out = torch.empty((B, C, Ho, Wo), device=dev, dtype=torch.float32)
    check(_L().gfn_grid_sample_fwd(ptr(x), ptr(g), ptr(out), C * Ho * Wo, B, C, H, W, Ho, Wo, stream_ptr(dev)),
          "gfn_grid_sample_fwd")
    return out


def interpolate_bilinear(x, size):
    """F.interpolate(x, size=size, mode='bilinear', align_corners=False) (model/network.py:238-249,271-281)."""
    dev = require_gpu(x)
    x = f32c(x)
    B, C, H, W = x.shape
    Ho, Wo = (int(size), int(size)) if isinstance(size, int) else (int(size[0]), int(size[1]))
    out = torch.empty((B, C, Ho, Wo), device=dev, dtype=torch.float32)
    check(_L().gfn_interp_bilinear_fwd(ptr(x), ptr(out), B * C, H, W, Ho, Wo, stream_ptr(dev)), "gfn_interp_bilinear_fwd")
    return out


def interpolate_bilinear_pair(a, b, size):
    """interpolate_bilinear of two tensors with the same batch and spatial size in one launch (flow + certainty,
    model/network.py:238-249,271-281)."""
    dev = require_gpu(a, b)
    a, b = f32c(a), f32c(b)
    B, Ca, H, W = a.shape
    if b.shape[0] != B or tuple(b.shape[2:]) != (H, W):
        raise ValueError(f"interpolate_bilinear_pair: {tuple(a.shape)} vs {tuple(b.shape)}")
    Cb = b.shape[1]
    Ho, Wo = (int(size), int(size)) if isinstance(size, int) else (int(size[0]), int(size[1]))
    oa = torch.empty((B, Ca, Ho, Wo), device=dev, dtype=torch.float32)
    ob = torch.empty((B, Cb, Ho, Wo), device=dev, dtype=torch.float32)
    check(_L().gfn_interp_bilinear_pair_fwd(ptr(a), ptr(oa), B * Ca, ptr(b), ptr(ob), B * Cb, H, W, Ho, Wo, stream_ptr(dev)),
          "gfn_interp_bilinear_pair_fwd")
    return oa, ob


def flow_update_(flow, certainty, delta, disp_prev, scale, W0, H0, zero_small=True, first_iteration=True):
    """In place: model/network.py:262-268.  delta is the refiner output (B,3,G,G) (channels 0,1 =
    displacement, 2 = certainty increment); disp_prev (B,2,G,G) carries the previous displacement."""
    dev = require_gpu(flow, certainty, delta, disp_prev)
    B, _, G, _ = flow.shape
    if tuple(certainty.shape) != (B, 1, G, G) or tuple(disp_prev.shape) != (B, 2, G, G) or \
            tuple(delta.shape[:1] + delta.shape[2:]) != (B, G, G) or delta.shape[1] < 3:
        raise ValueError("flow_update_: inconsistent shapes")
    for t in (flow, certainty, disp_prev):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError("flow_update_: flow/certainty/disp_prev must be contiguous fp32 (updated in place)")
    dl = f32c(delta)
    check(_L().gfn_flow_update_fwd(ptr(flow), ptr(certainty), ptr(dl), dl.shape[1] * G * G, ptr(disp_prev), B, G, int(scale),
                                   int(W0), int(H0), 1 if zero_small else 0, 1 if first_iteration else 0, stream_ptr(dev)),
          "gfn_flow_update_fwd")
    return flow, certainty


def _plane_view(t, planes, G):
    """(tensor, batch stride) of a (B, >=planes, G, G) fp32 tensor whose planes are contiguous (a channel slice is fine)."""
    if t.dtype == torch.float32 and t.stride(3) == 1 and t.stride(2) == G and t.stride(1) == G * G:
        return t, (t.stride(0) if t.shape[0] > 1 else t.shape[1] * G * G)
    t = f32c(t)
    return t, t.shape[1] * G * G


def flow_update(flow, certainty, d_flow, d_cert, disp_prev, scale, W0, H0, zero_small=True, first_iteration=True):
    """model/network.py:262-268 out of place: returns (flow + displacement(d_flow), certainty + d_cert) as new tensors (the
    reference keeps every iteration's result); d_flow (B,2,G,G) / d_cert (B,1,G,G) may be channel slices of one tensor.
    disp_prev=None (first iteration of a scale that has only one): the displacement is not stored."""
    dev = require_gpu(flow, certainty, d_flow, d_cert, *(() if disp_prev is None else (disp_prev,)))
    B, _, G, _ = flow.shape
    if tuple(certainty.shape) != (B, 1, G, G) or tuple(d_flow.shape) != (B, 2, G, G) or tuple(d_cert.shape) != (B, 1, G, G) or \
            (disp_prev is not None and tuple(disp_prev.shape) != (B, 2, G, G)):
        raise ValueError("flow_update: inconsistent shapes")
    if disp_prev is None:
        if not first_iteration:
            raise ValueError("flow_update: disp_prev=None is only valid for the first (and only) iteration of a scale")
    elif disp_prev.dtype != torch.float32 or not disp_prev.is_contiguous():
        raise ValueError("flow_update: disp_prev must be contiguous fp32 (updated in place)")
    fi, ci = f32c(flow), f32c(certainty)
    df, df_bs = _plane_view(d_flow, 2, G)
    dc, dc_bs = _plane_view(d_cert, 1, G)
    fo, co = torch.empty_like(fi), torch.empty_like(ci)
    check(_L().gfn_flow_update_out_fwd(ptr(fi), ptr(ci), ptr(fo), ptr(co), c_vp(df.data_ptr()), df_bs, c_vp(dc.data_ptr()), dc_bs,
                                       ptr(disp_prev), B, G, int(scale), int(W0), int(H0), 1 if zero_small else 0,
                                       1 if first_iteration else 0, stream_ptr(dev)), "gfn_flow_update_out_fwd")
    return fo, co


def match_post(flow, certainty, cert16=None, symmetric=True):
    """model/network.py:332-338 + 358-384: returns warp (B,G,2G,4)/(B,G,G,4) and certainty (B,G,2G)/(B,G,G)."""
    dev = require_gpu(flow, certainty, cert16)
    fl, ce = f32c(flow), f32c(certainty)
    nb, _, G, _ = fl.shape
    if tuple(ce.shape) != (nb, 1, G, G) or (cert16 is not None and cert16.shape[0] != nb) or (symmetric and nb % 2):
        raise ValueError("match_post: inconsistent shapes")
    B = nb // 2 if symmetric else nb
    Gw = 2 * G if symmetric else G
    c16 = f32c(cert16) if cert16 is not None else None
    Gc = c16.shape[-1] if c16 is not None else 0
    warp = torch.empty((B, G, Gw, 4), device=dev, dtype=torch.float32)
    cout = torch.empty((B, G, Gw), device=dev, dtype=torch.float32)
    check(_L().gfn_match_post_fwd(ptr(fl), ptr(ce), ptr(c16), ptr(warp), ptr(cout), B, G, Gc, 1 if symmetric else 0,
                                  stream_ptr(dev)), "gfn_match_post_fwd")
    return warp, cout


def kde_density(x, y=None, std=0.1, y_row_stride=None, cull=None, round_fp16=False):
    """sum_m exp(-|x_n - y_m|^2/(2 std^2)); x (N,D) or (Bt,N,D); y defaults to x.  fp32.
    round_fp16: coordinates rounded to fp16 first (what GFNet.sample hands to kde(); sums stay fp32).
    cull (default: automatic for 4-D points, N >= 4096): sort the points along a Morton curve of the
    A-image coordinates and skip blocks of reference points beyond 6.7 std (terms < 2^-32)."""
    dev = require_gpu(x, y)
    xs = f32c(x)
    squeeze = xs.dim() == 2
    if squeeze:
        xs = xs[None]
    Bt, N, D = xs.shape
    if y is None:
        ys, M, rs, bs = xs, N, D, N * D
    else:
        ys = f32c(y)
        if ys.dim() == 2:
            ys = ys[None]
        M, rs, bs = ys.shape[1], D, ys.shape[1] * D
    if y_row_stride is not None:  # strided view of ys (x[::down]) without a copy
        rs = int(y_row_stride)
        M = (ys.shape[1] * D + rs - 1) // rs
    if cull is None:
        cull = D == 4 and N >= 4096 and M >= 4096 and std <= 0.2
    if cull and D == 4:
        if y_row_stride is not None:
            ys = ys[:, ::rs // D].contiguous()
        same = y is None and y_row_stride is None
        out = _kde_culled(xs, xs if same else ys, std, same, dev, round_fp16)
        return out[0] if squeeze else out
    if round_fp16:  # only the culled path rounds on the device
        same_t = ys is xs
        xs = xs.half().float()
        ys = xs if same_t else ys.half().float()
    out = torch.empty((Bt, N), device=dev, dtype=torch.float32)
    nscr = int(_L().gfn_kde_scratch_floats(Bt, N, M, D))
    scratch = torch.empty((max(nscr, 4),), device=dev, dtype=torch.float32)
    check(_L().gfn_kde_density(ptr(xs), ptr(ys), ptr(out), Bt, N, M, D, rs, bs, float(std), ptr(scratch), nscr,
                               stream_ptr(dev)), "gfn_kde_density")
    return out[0] if squeeze else out


def _morton_sorted(pts, dev, long_perm=True):
    """(sorted points, permutation): rows stably ordered by the Morton key of their A-image position (one HIP launch:
    in-LDS two-pass radix sort per row, the same permutation as torch.sort(keys, stable=True))."""
    Bt, N, _ = pts.shape
    pts = f32c(pts)
    out = torch.empty_like(pts)
    perm = torch.empty((Bt, N), device=dev, dtype=torch.int32)
    tmp = torch.empty((Bt, N), device=dev, dtype=torch.int32)
    check(_L().gfn_kde_morton_sort(ptr(pts), ptr(out), ptr(perm), ptr(tmp), Bt, N, stream_ptr(dev)), "gfn_kde_morton_sort")
    return out, (perm.long() if long_perm else perm)


def _kde_culled(xs, ys, std, same, dev, round_fp16=False):
    Bt, N, _ = xs.shape
    M = ys.shape[1]
    xsort, perm = _morton_sorted(xs, dev, long_perm=False)
    ysort = xsort if same else _morton_sorted(ys, dev, long_perm=False)[0]
    out = torch.empty((Bt, N), device=dev, dtype=torch.float32)
    nscr = int(_L().gfn_kde_sorted_scratch_floats(Bt, N, M))
    scratch = torch.empty((nscr,), device=dev, dtype=torch.float32)
    # perm: the densities are written straight back in the caller's order
    check(_L().gfn_kde_density_sorted(ptr(xsort), ptr(ysort), ptr(out), ptr(perm), Bt, N, M, float(std), 1 if round_fp16 else 0, ptr(scratch), nscr,
                                      stream_ptr(dev)), "gfn_kde_density_sorted")
    return out


def threshold_certainty(certainty, thresh):
    """certainty[certainty > thresh] = 1 on a copy (model/network.py:391-393)."""
    dev = require_gpu(certainty)
    c = f32c(certainty)
    out = torch.empty_like(c)
    check(_L().gfn_threshold_certainty(ptr(c), ptr(out), c.numel(), float(thresh), stream_ptr(dev)), "gfn_threshold_certainty")
    return out


def balance_weights(density, min_density=10.0, floor_p=1e-7, round_fp16=False):
    """p = 1/(density+1); p[density < 10] = 1e-7 (model/network.py:409-410).  round_fp16: the density is rounded to fp16
    first, as kde(half=True) returns it."""
    dev = require_gpu(density)
    d = f32c(density)
    p = torch.empty_like(d)
    check(_L().gfn_balance_weights(ptr(d), ptr(p), d.numel(), float(min_density), float(floor_p), 1 if round_fp16 else 0, stream_ptr(dev)),
          "gfn_balance_weights")
    return p


def gather_matches(matches, certainty, idx, one_above=None):
    """(matches[b, idx[b]], certainty[b, idx[b]]) for a batch (model/network.py:403-404, 414) in one launch; one_above:
    certainties above it come back as 1 (the threshold of network.py:391-393 applied on the fly)."""
    dev = require_gpu(matches, certainty, idx)
    m, c = f32c(matches), f32c(certainty)
    Bt, N, four = m.shape
    K = idx.shape[1]
    if four != 4 or tuple(c.shape) != (Bt, N) or idx.shape[0] != Bt or idx.dtype != torch.int64:
        raise ValueError("gather_matches: matches (Bt,N,4), certainty (Bt,N), idx (Bt,K) int64")
    idx = idx.contiguous()
    om = torch.empty((Bt, K, 4), device=dev, dtype=torch.float32)
    oc = torch.empty((Bt, K), device=dev, dtype=torch.float32)
    check(_L().gfn_gather_matches(ptr(m), ptr(c), ptr(idx), ptr(om), ptr(oc), Bt, N, K,
                                  float("inf") if one_above is None else float(one_above), stream_ptr(dev)), "gfn_gather_matches")
    return om, oc


def sample_without_replacement(weights, num_samples, seed=None, one_above=None):
    """torch.multinomial(weights, num_samples, replacement=False) for a (Bt,N) batch (model/network.py:400-402, 411-413):
    exponential race, one HIP launch pair; indices come back in increasing order.  seed=None draws one from torch's
    default CPU generator, so torch.manual_seed() makes runs repeatable.  one_above: weights above it count as 1."""
    dev = require_gpu(weights)
    w = f32c(weights)
    if w.dim() != 2:
        raise ValueError("sample_without_replacement: weights must be (Bt, N)")
    Bt, N = w.shape
    K = int(num_samples)
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    out = torch.empty((Bt, K), device=dev, dtype=torch.int64)
    scratch = torch.empty((Bt * (N + 2048),), device=dev, dtype=torch.int32)
    check(_L().gfn_sample_without_replacement(ptr(w), N, ptr(out), ptr(scratch), Bt, N, K, int(seed) & (2 ** 64 - 1),
                                              float("inf") if one_above is None else float(one_above), stream_ptr(dev)),
          "gfn_sample_without_replacement")
    return out


def convert_matches(matches, wA, hA, wB, hB):
    """estimation.py:26-45 on the device: (...,4) normalised warp rows -> pixel (x,y,u,v), float32."""
    dev = require_gpu(matches)
    m = f32c(matches)
    out = torch.empty_like(m)
    n = m.numel() // 4
    check(_L().gfn_convert_matches(ptr(m), ptr(out), n, float(wA), float(hA), float(wB), float(hB), stream_ptr(dev)),
          "gfn_convert_matches")
    return out


def find_homography(pts, thresh=3.0, iters=2000, seed=0, lm_iters=10, stage=0, return_mask=False, confidence=0.99999,
                    return_iters=False):
    """Batched stand-in for cv2.findHomography(pos_a, pos_b, cv2.RANSAC, confidence=0.99999, ransacReprojThreshold=thresh)
    (estimation.py:66-72), on the device.  pts (Bt,N,4) or (N,4) pixel (x,y,u,v).  confidence: OpenCV's termination rule
    (the iteration bound follows the best inlier ratio; `iters` = maxIters); 0 scores all `iters` hypotheses.
    Returns H (Bt,3,3) float64, inlier counts (Bt,), chosen hypothesis index (Bt,) [, mask (Bt,N) uint8] [, iteration bound at
    exit (Bt,)]."""
    dev = require_gpu(pts)
    p = f32c(pts)
    if p.dim() == 2:
        p = p[None]
    Bt, N, _ = p.shape
    H = torch.empty((Bt, 3, 3), device=dev, dtype=torch.float64)
    ninl = torch.empty((Bt,), device=dev, dtype=torch.int32)
    best = torch.empty((Bt,), device=dev, dtype=torch.int32)
    used = torch.empty((Bt,), device=dev, dtype=torch.int32) if return_iters else None
    mask = torch.empty((Bt, N), device=dev, dtype=torch.uint8) if return_mask else None
    nb = int(_L().gfn_homography_scratch_bytes(Bt, int(iters)))
    scratch = torch.empty((nb // 8 + 1,), device=dev, dtype=torch.float64)
    check(_L().gfn_homography_ransac_ex(ptr(p), Bt, N, float(thresh), int(iters), float(confidence or 0.0), int(seed), int(lm_iters),
                                        int(stage), ptr(H), ptr(ninl), ptr(best), ptr(mask), ptr(used), ptr(scratch), nb, stream_ptr(dev)),
          "gfn_homography_ransac")
    out = (H, ninl, best)
    if return_mask:
        out = out + (mask,)
    return out + (used,) if return_iters else out


def homography_dlt(pts, weight=None):
    """One-shot weighted normalised DLT over all correspondences ("grid-DLT").  pts (Bt,N,4) pixels,
    weight (Bt,N) or None -> H (Bt,3,3) float64, ok (Bt,) int32."""
    dev = require_gpu(pts, weight)
    p = f32c(pts)
    if p.dim() == 2:
        p = p[None]
    Bt, N, _ = p.shape
    w = f32c(weight).reshape(Bt, N) if weight is not None else None
    H = torch.empty((Bt, 3, 3), device=dev, dtype=torch.float64)
    ok = torch.empty((Bt,), device=dev, dtype=torch.int32)
    check(_L().gfn_homography_dlt(ptr(p), ptr(w), Bt, N, ptr(H), ptr(ok), stream_ptr(dev)), "gfn_homography_dlt")
    return H, ok


IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def resize_normalise(im, size, mode="bicubic", mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """get_tuple_transform_ops(resize=size, mode, normalize=True) of the reference (utils/utils.py:18-27) on a
    (B,>=3,H,W) float image batch in [0,1]: F.interpolate(mode, align_corners=False, antialias=False) + (x-mean)/std on
    the first three channels, one HIP kernel.  mode: 'bicubic' or 'bilinear' (the reference's mode=2)."""
    import ctypes

    dev = require_gpu(im)
    if im.dim() != 4 or im.shape[1] < 3:
        raise ValueError("resize_normalise: expected (B, >=3, H, W)")
    if mode not in ("bilinear", "bicubic"):
        raise ValueError("resize_normalise: mode must be 'bilinear' or 'bicubic'")
    x = f32c(im)
    B, C, H, W = x.shape
    Ho, Wo = (int(size), int(size)) if isinstance(size, int) else (int(size[0]), int(size[1]))
    out = torch.empty((B, 3, Ho, Wo), device=dev, dtype=torch.float32)
    m3, s3 = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    check(_L().gfn_resize_normalize_fwd(ptr(x), C * H * W, ptr(out), B, H, W, Ho, Wo, 1 if mode == "bicubic" else 0, m3, s3,
                                        stream_ptr(dev)), "gfn_resize_normalize_fwd")
    return out


def conv_block_pack(dw_w, dw_b, bn_alpha, bn_beta, pw_w, pw_b):
    """Pack one ConvRefiner block (model/network.py:471-487) for conv_block: dw_w (C,25) or (C,1,5,5),
    dw_b (C) or None, eval-mode BatchNorm as y = x*alpha + beta, pw_w (M,C[,1,1]), pw_b (M)."""
    dev = require_gpu(dw_w, pw_w)
    C, M = dw_w.shape[0], pw_w.shape[0]
    dw_w, pw_w = f32c(dw_w.reshape(C, 25)), f32c(pw_w.reshape(M, C))
    al, be, pb = f32c(bn_alpha), f32c(bn_beta), f32c(pw_b)
    db = f32c(dw_b) if dw_b is not None else None
    packed = torch.empty(int(_L().gfn_conv_block_packed_floats(C, M)), device=dev, dtype=torch.float32)
    check(_L().gfn_conv_block_pack(ptr(dw_w), ptr(db) if db is not None else None, ptr(al), ptr(be), ptr(pw_w), ptr(pb), ptr(packed),
                                   C, M, stream_ptr(dev)), "gfn_conv_block_pack")
    return packed


def conv_block(x, packed, M, out=None, variant=0, t_scratch=None):
    """Conv2d(C,C,5,pad 2,groups=C) -> BatchNorm2d(eval) -> ReLU -> Conv2d(C,M,1) in one kernel
    (model/network.py:471-487).  variant bit 0: the two-pass form (bit-identical to the fused one);
    bit 1: 1x1 conv with fp16 operands (W and the ReLU output rounded to fp16, fp32 accumulation) --
    the reference's autocast numerics class (amp=True refiners) -- instead of fp32 throughout."""
    dev = require_gpu(x, packed)
    x = f32c(x)
    B, C, G, G2 = x.shape
    if G != G2:
        raise ValueError("conv_block: square grids only")
    if packed.numel() != int(_L().gfn_conv_block_packed_floats(C, M)):
        raise ValueError("conv_block: packed parameters do not match (C=%d, M=%d)" % (C, M))
    if out is None:
        out = torch.empty((B, M, G, G), device=dev, dtype=torch.float32)
    if ((variant & 1) or G % 4) and t_scratch is None:
        t_scratch = torch.empty_like(x)
    check(_timed("conv_block_c%d_g%d" % (C, G), lambda: _L().gfn_conv_block_fwd(
        ptr(x), ptr(packed), ptr(out), ptr(t_scratch) if t_scratch is not None else None, B, C, M, G, int(variant),
        stream_ptr(dev))), "gfn_conv_block_fwd")
    return out


def conv_block_half(x, packed, C, M, out=None, out_half=True):
    """conv_block on fp16 maps (the reference's amp=True class, model/network.py:560-562): x is (B,C,G,G) float32 or a
    half map (B,ceil(C/2),G,G,2) float16 (channel pairs side by side); returns a half map (B,ceil(M/2),G,G,2), or (B,M,G,G)
    float32 with out_half=False.  The autocast class: depthwise 5x5 and 1x1 operands fp16 (the input halo, the folded taps, the
    ReLU output and the 1x1 weights are rounded), every accumulation, BatchNorm and ReLU fp32 (include/gfnet_hip.h)."""
    dev = require_gpu(x, packed)
    x_half = x.dtype == torch.float16
    if x_half:
        if x.dim() != 5 or x.shape[1] != (C + 1) // 2 or x.shape[4] != 2 or not x.is_contiguous():
            raise ValueError("conv_block_half: a half map is a contiguous (B, ceil(C/2), G, G, 2) float16 tensor")
    else:
        x = f32c(x)
        if x.dim() != 4 or x.shape[1] != C:
            raise ValueError("conv_block_half: x must be (B, C, G, G)")
    B, G, G2 = x.shape[0], x.shape[2], x.shape[3]
    if G != G2:
        raise ValueError("conv_block_half: square grids only")
    if packed.numel() != int(_L().gfn_conv_block_packed_floats(C, M)):
        raise ValueError("conv_block_half: packed parameters do not match (C=%d, M=%d)" % (C, M))
    shape = (B, (M + 1) // 2, G, G, 2) if out_half else (B, M, G, G)
    dt = torch.float16 if out_half else torch.float32
    if out is None or tuple(out.shape) != shape or out.dtype != dt:
        out = torch.empty(shape, device=dev, dtype=dt)
    check(_timed("conv_block_half_c%d_g%d" % (C, G), lambda: _L().gfn_conv_block_half_fwd(
        ptr(x), _lib.GFN_F16 if x_half else _lib.GFN_F32, ptr(packed), ptr(out), _lib.GFN_F16 if out_half else _lib.GFN_F32, B, C, M, G,
        stream_ptr(dev))), "gfn_conv_block_half_fwd")
    return out


def half_map_to_float(h, C):
    """(B, ceil(C/2), G, G, 2) float16 half map -> (B, C, G, G) float32 (tests, debugging)."""
    B, NP, G, G2, _ = h.shape
    return h.permute(0, 1, 4, 2, 3).reshape(B, 2 * NP, G, G2)[:, :C].float()


def pointwise_conv(t, w, bias, out=None):
    """Conv2d(K, M, 1)(t) for a few output channels: out_conv (model/network.py:505,563).  w (M,K)."""
    dev = require_gpu(t, w)
    t, w, bias = f32c(t), f32c(w), f32c(bias)
    B, K, G, G2 = t.shape
    M = bias.shape[0]
    if out is None:
        out = torch.empty((B, M, G, G2), device=dev, dtype=torch.float32)
    check(_timed("pw_m%d_k%d_g%d" % (M, K, G), lambda: _L().gfn_pointwise_conv_fwd(
        ptr(w), ptr(bias), ptr(t), ptr(out), B, M, K, G * G2, stream_ptr(dev))), "gfn_pointwise_conv_fwd")
    return out
