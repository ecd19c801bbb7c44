"""Drop-in for the reference's utils/local_correlation.py (same name, same signature).

`local_correlation(...)` keeps the argument list of utils/local_correlation.py:4-16 so that
model/network.py:553-554 can import it unchanged; the arithmetic runs in the hand-written gfx950
kernels of csrc/local_corr.hip through the C ABI (gfn_local_corr_fwd, include/gfnet_hip.h).
"""
import torch

from .. import _lib


def _channel_stride_view(t):
    """Return (tensor, batch_stride) for a (B,C,G,G) tensor whose channel/spatial dims are
    contiguous but whose batch stride may be larger (a channel slice of a concat buffer)."""
    B, C, G1, G2 = t.shape
    if t.dtype == torch.float32 and t.stride(3) == 1 and t.stride(2) == G2 and t.stride(1) == G1 * G2 and \
            (B == 1 or t.stride(0) >= C * G1 * G2):
        return t, (t.stride(0) if B > 1 else C * G1 * G2)
    t = _lib.f32c(t)
    return t, C * G1 * G2


def local_correlation(featuremap_size, feature0, feature1, local_radius, num_grid, padding_mode="zeros", flow=None,
                      im_A_coords=None, sample_mode="bilinear", grid_based_correlation=False, num_level=1, out=None,
                      _variant=0):
    """Local (2r+1)^2 correlation of grid features against feature1 sampled around `flow`.

    Same contract as the reference (utils/local_correlation.py:4-72): returns (B, K*num_level,
    num_grid, num_grid) with dtype/device of feature0; `im_A_coords` is accepted and ignored.
    Extra keyword `out`: a (B, K*num_level, G, G) fp32 view to write into (e.g. the channel slice
    of the refiner's concat buffer); it must have contiguous (K,G,G) planes.
    Gradients: like the reference (local_correlation.py:54-60, sampling under no_grad) only feature0
    receives one; it is computed by gfn_local_corr_bwd_f0 when feature0.requires_grad (and `out` is None).
    """
    if out is None and torch.is_grad_enabled() and feature0.requires_grad:
        return _LocalCorrelationFn.apply(feature0, feature1, flow, tuple(int(v) for v in featuremap_size), int(local_radius),
                                         int(num_grid), bool(grid_based_correlation), int(num_level))
    return _forward(featuremap_size, feature0, feature1, local_radius, num_grid, padding_mode, flow, sample_mode,
                    grid_based_correlation, num_level, out, _variant)


def _forward(featuremap_size, feature0, feature1, local_radius, num_grid, padding_mode, flow, sample_mode,
             grid_based_correlation, num_level, out, _variant):
    if padding_mode != "zeros" or sample_mode != "bilinear":
        raise ValueError("only padding_mode='zeros', sample_mode='bilinear' (the reference's settings) are supported")
    B, c, h, w = [int(v) for v in featuremap_size]
    r = int(local_radius)
    G = int(num_grid)
    dev = _lib.require_gpu(feature0, feature1, flow)
    K1 = (2 * r + 1) ** 2
    K = K1 * int(num_level)
    if tuple(feature0.shape) != (B, c, G, G):
        raise ValueError(f"feature0 must be (B,c,num_grid,num_grid)={(B, c, G, G)}, got {tuple(feature0.shape)}")
    if tuple(feature1.shape) != (B, c, h, w):
        raise ValueError(f"feature1 must match featuremap_size {(B, c, h, w)}, got {tuple(feature1.shape)}")
    if flow is not None and tuple(flow.shape) != (B, 2, G, G):
        raise ValueError(f"flow must be (B,2,num_grid,num_grid), got {tuple(flow.shape)}")
    if flow is None and not (G == h == w):
        raise ValueError("flow=None assumes aligned maps: num_grid == h == w")
    f0, f0_bs = _channel_stride_view(feature0.detach())
    f1, f1_dt = _lib.featc(feature1.detach())  # fp16 maps are read as stored
    fl = _lib.f32c(flow.detach()) if flow is not None else None
    ret_dtype = feature0.dtype
    if out is None:
        res = torch.empty((B, K, G, G), device=dev, dtype=torch.float32)
        out_bs = K * G * G
    else:
        if tuple(out.shape) != (B, K, G, G) or out.dtype != torch.float32 or out.stride(3) != 1 or \
                out.stride(2) != G or out.stride(1) != G * G:
            raise ValueError("out must be a fp32 (B,K,G,G) view with contiguous (K,G,G) planes")
        res = out
        out_bs = out.stride(0) if B > 1 else K * G * G
    L = _lib.lib()
    st = _lib.stream_ptr(dev)
    nscr = int(L.gfn_local_corr_scratch_bytes(B, G))
    scr = _lib.scratch(dev, nscr)
    hh, ww = h, w
    for level in range(int(num_level)):
        o = res[:, level * K1:(level + 1) * K1]
        from .. import ops  # (LOCAL_CORR_FP32: fp32 FMA arithmetic at every radius instead of the matrix-core kernel for r >= 5)
        variant = 4 if (int(_variant) == 0 and ops.LOCAL_CORR_FP32) else int(_variant)
        _lib.check(L.gfn_local_corr_fwd_dt(_lib.ptr(f0), f0_bs, _lib.ptr(f1), None, f1_dt, _lib.ptr(fl), _lib.c_vp(o.data_ptr()),
                                           out_bs, B, c, G, hh, ww, r, 1 if grid_based_correlation else 0, h, w,
                                           variant, _lib.ptr(scr), nscr, st), "gfn_local_corr_fwd")
        if level + 1 < num_level:
            if f1_dt != _lib.GFN_F32:  # pooled levels (unused by GFNet) are built in fp32
                f1, f1_dt = _lib.f32c(f1), _lib.GFN_F32
            pooled = torch.empty((B, c, hh // 2, ww // 2), device=dev, dtype=torch.float32)
            _lib.check(L.gfn_avg_pool2(_lib.ptr(f1), _lib.ptr(pooled), B * c, hh, ww, st), "gfn_avg_pool2")
            f1, hh, ww = pooled, hh // 2, ww // 2
    if out is None and ret_dtype != torch.float32:
        res = res.to(ret_dtype)
    return res


def _pyramid(f1, B, c, h, w, num_level, dev):
    """feature1 and its 2x average-pooled levels (local_correlation.py:71), as the forward builds them."""
    L = _lib.lib()
    st = _lib.stream_ptr(dev)
    levels, hh, ww = [(f1, h, w)], h, w
    for _ in range(1, num_level):
        pooled = torch.empty((B, c, hh // 2, ww // 2), device=dev, dtype=torch.float32)
        _lib.check(L.gfn_avg_pool2(_lib.ptr(f1), _lib.ptr(pooled), B * c, hh, ww, st), "gfn_avg_pool2")
        f1, hh, ww = pooled, hh // 2, ww // 2
        levels.append((f1, hh, ww))
    return levels


class _LocalCorrelationFn(torch.autograd.Function):
    """local_correlation with the reference's gradient: d/d feature0 only."""

    @staticmethod
    def forward(ctx, feature0, feature1, flow, featuremap_size, r, G, grid_based, num_level):
        res = _forward(featuremap_size, feature0, feature1, r, G, "zeros", flow, "bilinear", grid_based, num_level, None, 0)
        ctx.save_for_backward(feature1, flow if flow is not None else torch.empty(0, device=feature0.device))
        ctx.meta = (featuremap_size, r, G, grid_based, num_level, flow is not None, feature0.dtype)
        return res

    @staticmethod
    def backward(ctx, grad_out):
        feature1, flow = ctx.saved_tensors
        (B, c, h, w), r, G, grid_based, num_level, has_flow, dtype = ctx.meta
        dev = grad_out.device
        K1 = (2 * r + 1) ** 2
        g = _lib.f32c(grad_out)
        fl = _lib.f32c(flow) if has_flow else None
        L = _lib.lib()
        st = _lib.stream_ptr(dev)
        total = None
        for level, (f1, hh, ww) in enumerate(_pyramid(_lib.f32c(feature1), B, c, h, w, num_level, dev)):
            gl = g[:, level * K1:(level + 1) * K1]
            gf0 = torch.empty((B, c, G, G), device=dev, dtype=torch.float32)
            _lib.check(L.gfn_local_corr_bwd_f0(_lib.c_vp(gl.data_ptr()), g.stride(0), _lib.ptr(f1), None, _lib.ptr(fl), _lib.ptr(gf0),
                                               c * G * G, B, c, G, hh, ww, r, 1 if grid_based else 0, h, w, st), "gfn_local_corr_bwd_f0")
            total = gf0 if total is None else total + gf0
        return total.to(dtype), None, None, None, None, None, None, None
