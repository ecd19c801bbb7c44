/*
 * gfnet_hip.h -- C ABI of libgfnet_hip.so: the MI355X (gfx950) kernels behind GFNet's
 * dense-correlation -> flow -> balanced-sampling -> homography-solve path.
 *
 * The reference (KN-Zhang/GFNet) has no FFI layer: its boundary is a set of Python functions.
 * Each entry point below replaces the arithmetic of one of them; the Python mirror that keeps the
 * reference's signatures lives in gfnet_amd/ (see INTEGRATION.md for the binding a maintainer of
 * the reference would add).  Paths cited are relative to the reference repository.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; every tensor pointer is a DEVICE pointer to fp32 data,
 *     NCHW-contiguous unless a stride argument says otherwise; the caller owns every buffer,
 *     the library never allocates or frees device memory (scratch is passed in explicitly);
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the default stream) and
 *     the call returns without synchronising; it is safe to capture into a hipGraph;
 *   - return value: GFN_OK (0) or a negative GFN_ERR_* code; gfn_last_error() gives the text
 *     (thread-local).  Nothing is launched when an argument error is returned;
 *   - no global mutable state; single host thread per stream is assumed, as in the reference.
 */
#ifndef GFNET_HIP_H
#define GFNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GFN_OK 0
#define GFN_ERR_INVALID_ARG (-1) /* bad size / null pointer / unsupported combination */
#define GFN_ERR_LAUNCH (-2)      /* hipLaunchKernel / runtime error, see gfn_last_error() */
#define GFN_ERR_SCRATCH (-3)     /* scratch buffer too small */

#define GFN_ABI_VERSION 1

typedef void *gfn_stream_t; /* hipStream_t */

/* Library / device introspection.  gfn_device_arch() copies the gcnArchName of the current
 * device ("gfx950...") into buf. */
int gfn_abi_version(void);
const char *gfn_last_error(void);
int gfn_device_arch(char *buf, int buflen);

/* ---------------------------------------------------------------------------------------------
 * local_correlation -- utils/local_correlation.py:4-72 (called from model/network.py:553-554).
 *
 *   out[b, k, i, j] = sum_c f0[b,c,i,j] / sqrt(C) * bilinear(f1[b,c], p(b,i,j) + offset(k))
 *
 * k = ky*(2r+1)+kx (x fastest); zeros padding, align_corners=False; p = flow[b,:,i,j] in
 * normalised [-1,1] coordinates, or the identity grid when flow == NULL (then G == win_h ==
 * win_w is required, as in the reference).  The window offsets are
 * linspace(-2r/win_h, 2r/win_h, 2r+1) x linspace(-2r/win_w, 2r/win_w, 2r+1) normalised units
 * (grid_based != 0: +-2r/G on both axes, local_correlation.py:34-40).  For num_level == 1 the
 * reference has win_h == H, win_w == W (one feature pixel per tap); on pooled levels
 * (local_correlation.py:61-71) the caller passes the pooled f1 with the original win_h/win_w.
 *
 *   f0   (B,C,G,G)  batch stride f0_bs floats (>= C*G*G; lets f0 live inside a concat buffer)
 *   f1   (B,C,H,W)  contiguous
 *   flow (B,2,G,G)  contiguous, or NULL
 *   out  (B,K,G,G)  batch stride out_bs floats (>= K*G*G), K = (2r+1)^2
 *
 * Fast path (LDS-tiled, shared bilinear fractions): C % 16 == 0, 1 <= r <= 7, !grid_based,
 * win_h == H, win_w == W.  Anything else runs the general per-tap kernel.  Flow values are
 * unrestricted (out-of-image taps read zeros); tiles whose search windows do not fit the LDS
 * stage fall back to the per-tap path inside the same launch.
 */
int gfn_local_corr_fwd(const float *f0, int64_t f0_bs, const float *f1, const float *flow, float *out, int64_t out_bs,
                       int B, int C, int G, int H, int W, int r, int grid_based, int win_h, int win_w,
                       gfn_stream_t stream);

/* Variant selector for experiments/tests: 0 = auto (as above), 1 = force the general per-tap
 * kernel.  Same arguments otherwise. */
int gfn_local_corr_fwd_ex(const float *f0, int64_t f0_bs, const float *f1, const float *flow, float *out,
                          int64_t out_bs, int B, int C, int G, int H, int W, int r, int grid_based, int win_h,
                          int win_w, int variant, gfn_stream_t stream);

/* F.avg_pool2d(x, 2, 2) between correlation levels (utils/local_correlation.py:71).
 * in (BC,H,W) -> out (BC,H/2,W/2). */
int gfn_avg_pool2(const float *in, float *out, int BC, int H, int W, gfn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GFNET_HIP_H */
