"""ctypes binding of libgfnet_hip.so (C ABI: include/gfnet_hip.h).

torch is imported first so that the library binds to the HIP runtime torch already loaded
(same libamdhip64 SONAME); torch itself is only plumbing here: device memory and streams.
There is NO CPU fallback: if the library is missing or the tensors are not on a GPU the ops raise.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GFNET_HIP_LIB") or os.path.join(_HERE, "csrc", "libgfnet_hip.so")
_lib = None

c_int, c_i64, c_vp, c_float, c_double = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_float, ctypes.c_double

# name -> argtypes; every entry point returns int (GFN_OK or a negative error code)
_SIGNATURES = {
    "gfn_local_corr_fwd": [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64] + [c_int] * 9 + [c_vp, c_i64, c_vp],
    "gfn_local_corr_fwd_ex": [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64] + [c_int] * 10 + [c_vp, c_i64, c_vp],
    "gfn_local_corr_fwd_dt": [c_vp, c_i64, c_vp, c_vp, c_int, c_vp, c_vp, c_i64] + [c_int] * 10 + [c_vp, c_i64, c_vp],
    "gfn_avg_pool2": [c_vp, c_vp, c_int, c_int, c_int, c_vp],
    "gfn_corr_softargmax_fwd": [c_vp, c_vp, c_vp] + [c_int] * 7 + [c_vp],
    "gfn_corr_volume_fwd": [c_vp, c_vp, c_vp, c_vp] + [c_int] * 6 + [c_vp],
    "gfn_pos_embed_fwd": [c_vp, c_vp] + [c_int] * 5 + [c_vp],
    "gfn_refiner_input_fwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64] + [c_int] * 6 + [c_float, c_int, c_vp],
    "gfn_refiner_input_fwd_dt": [c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_i64] + [c_int] * 6 + [c_float, c_int, c_vp],
    "gfn_refiner_input_plan_fwd_dt": [c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_i64] + [c_int] * 6 + [c_float, c_int, c_int, c_vp, c_i64, c_vp],
    "gfn_corr_softargmax_fwd_dt": [c_vp, c_vp, c_int, c_vp] + [c_int] * 7 + [c_vp],
    "gfn_corr_softargmax_fwd_ws": [c_vp, c_vp, c_int, c_vp] + [c_int] * 7 + [c_vp, c_i64, c_vp],
    "gfn_grid_sample_fwd": [c_vp, c_vp, c_vp, c_i64] + [c_int] * 6 + [c_vp],
    "gfn_interp_bilinear_fwd": [c_vp, c_vp] + [c_int] * 5 + [c_vp],
    "gfn_interp_bilinear_pair_fwd": [c_vp, c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp],
    "gfn_flow_update_fwd": [c_vp, c_vp, c_vp, c_i64, c_vp] + [c_int] * 7 + [c_vp],
    "gfn_flow_update_out_fwd": [c_vp] * 5 + [c_i64, c_vp, c_i64, c_vp] + [c_int] * 7 + [c_vp],
    "gfn_match_post_fwd": [c_vp, c_vp, c_vp, c_vp, c_vp] + [c_int] * 4 + [c_vp],
    "gfn_kde_msplit": [c_int, c_int, c_int],
    "gfn_kde_density": [c_vp, c_vp, c_vp] + [c_int] * 4 + [c_i64, c_i64, c_double, c_vp, c_i64, c_vp],
    "gfn_kde_morton_keys": [c_vp, c_vp, c_i64, c_vp],
    "gfn_kde_morton_sort": [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp],
    "gfn_kde_density_sorted": [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_double, c_int, c_vp, c_i64, c_vp],
    "gfn_threshold_certainty": [c_vp, c_vp, c_i64, c_float, c_vp],
    "gfn_balance_weights": [c_vp, c_vp, c_i64, c_float, c_float, c_int, c_vp],
    "gfn_sample_without_replacement": [c_vp, c_i64, c_vp, c_vp, c_int, c_int, c_int, ctypes.c_uint64, c_float, c_vp],
    "gfn_gather_matches": [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_float, c_vp],
    "gfn_convert_matches": [c_vp, c_vp, c_i64] + [c_float] * 4 + [c_vp],
    "gfn_homography_ransac": [c_vp, c_int, c_int, c_double, c_int, ctypes.c_uint64, c_int, c_int, c_vp, c_vp, c_vp, c_vp,
                              c_vp, c_i64, c_vp],
    "gfn_homography_ransac_ex": [c_vp, c_int, c_int, c_double, c_int, c_double, ctypes.c_uint64, c_int, c_int, c_vp, c_vp, c_vp, c_vp,
                                 c_vp, c_vp, c_i64, c_vp],
    "gfn_homography_dlt": [c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp],
    "gfn_local_corr_bwd_f0": [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64] + [c_int] * 9 + [c_vp],
    "gfn_resize_normalize_fwd": [c_vp, c_i64, c_vp] + [c_int] * 6 + [c_vp, c_vp, c_vp],
    "gfn_conv_block_pack": [c_vp] * 7 + [c_int] * 2 + [c_vp],
    "gfn_conv_block_fwd": [c_vp] * 4 + [c_int] * 5 + [c_vp],
    "gfn_conv_block_half_fwd": [c_vp, c_int, c_vp, c_vp] + [c_int] * 5 + [c_vp],
    "gfn_pointwise_conv_fwd": [c_vp] * 4 + [c_int] * 4 + [c_vp],
}
# entry points that return a size instead of a status
_SIZE_FUNCS = {
    "gfn_local_corr_scratch_bytes": [c_int, c_int],
    "gfn_corr_softargmax_ws_bytes": [c_int] * 4,
    "gfn_local_corr_plans": [c_int] * 6,
    "gfn_kde_scratch_floats": [c_int, c_int, c_int, c_int],
    "gfn_kde_sorted_scratch_floats": [c_int, c_int, c_int],
    "gfn_homography_scratch_bytes": [c_int, c_int],
    "gfn_conv_block_packed_floats": [c_int, c_int],
}


class GfnError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GfnError(f"{LIB_PATH} not found: build it with `python -m gfnet_amd.build` "
                           "(there is no CPU fallback for the GFNet hot path)")
        L = ctypes.CDLL(LIB_PATH)
        L.gfn_abi_version.restype = c_int
        L.gfn_last_error.restype = ctypes.c_char_p
        L.gfn_device_arch.argtypes = [ctypes.c_char_p, c_int]
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = c_int
        for name, argtypes in _SIZE_FUNCS.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = c_i64
        _lib = L
    return _lib


def exported_symbols():
    return ["gfn_abi_version", "gfn_last_error", "gfn_device_arch"] + list(_SIGNATURES) + list(_SIZE_FUNCS)


def check(code, what):
    if code != 0:
        msg = lib().gfn_last_error().decode()
        # gfn_local_corr_fwd wants its scratch counters zero on entry and only a call that ran to the end leaves them so
        # (include/gfnet_hip.h): after any failure the cached buffers are dropped, the next call gets freshly zeroed ones
        for b in _scratch.values():
            _retire(b)  # (not freed where a captured graph may still name them)
        _scratch.clear()
        raise GfnError(f"{what} failed ({code}): {msg}")


def ptr(t):
    return c_vp(t.data_ptr()) if t is not None else c_vp(0)


# the current stream's raw handle: torch.cuda.current_stream() builds a Stream object per call (~4.5 us; a step makes ~150 calls and the
# three-scene workload is bound by the host's launch rate), the C hook returns the handle itself
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def current_stream_handle(device):
    if _raw_stream is not None:
        idx = device.index
        return _raw_stream(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(device).cuda_stream


def stream_ptr(device):
    return c_vp(current_stream_handle(device))


def require_gpu(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise GfnError("gfnet_amd ops need tensors on an AMD GPU (torch device 'cuda'); there is no CPU path")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise GfnError(f"tensors on different devices: {dev} vs {t.device}")
    return dev


_scratch = {}
_retired = []  # outgrown or dropped scratch buffers that were handed out DURING a stream capture: a hipGraph still names them
_in_graphs = set()  # data_ptr of every buffer scratch() returned while its stream was capturing


def _retire(buf):
    """A buffer leaves the cache: kept alive only if a captured graph may name it (the list used to take every outgrown or
    dropped buffer, ~9 MB each at 448b32, whether or not a graph had ever been captured)."""
    if buf.data_ptr() in _in_graphs:
        _retired.append(buf)


def release_retired():
    """Free the scratch buffers kept alive for captured graphs; call after destroying those graphs."""
    _in_graphs.difference_update(b.data_ptr() for b in _retired)
    _retired.clear()


def scratch(device, nbytes):
    """A per-(device, stream), grow-only int32 scratch buffer.  Reuse is stream-ordered: any number of models may share it on
    one stream (their launches cannot overlap), work on different streams gets different buffers; one host thread per
    stream.  Dropped as a whole after any failed call (check())."""
    key = (device.type, device.index, current_stream_handle(device) if device.type == "cuda" else 0)
    buf = _scratch.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        # a buffer that is outgrown is RETIRED, not freed: a hipGraph captured on this stream keeps launching kernels with its
        # address (round 4: two scenes of different sizes captured on one stream -- the second capture grew the buffer and the first
        # graph's replays ended in a memory fault).  A few MB per growth step, a handful of steps per process.
        if buf is not None:
            _retire(buf)
        # zero-filled: gfn_local_corr_fwd wants its counters zero on entry and leaves them zero (include/gfnet_hip.h)
        buf = torch.zeros((max(nbytes, 1 << 16) + 3) // 4, device=device, dtype=torch.int32)
        _scratch[key] = buf
    if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
        _in_graphs.add(buf.data_ptr())
    return buf


GFN_F32, GFN_F16 = 0, 1


def featc(t):
    """A feature map as the kernels read it: contiguous, fp32 or fp16 as stored (BASELINE config 5: fp16 pyramids are read
    directly, no widened copy); returns (tensor, dtype code).  Other dtypes (bf16, fp64) are widened to fp32."""
    if t.dtype == torch.float16:
        return t.contiguous(), GFN_F16
    return f32c(t), GFN_F32


def f32c(t):
    """fp32 + contiguous (no copy when already so)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()
