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

/* storage type of the feature pyramids (BASELINE config 5 keeps them in fp16; reference: model/network.py:171-173 runs the
 * backbone under autocast, utils/utils.py:306-320).  Arithmetic and every other tensor stay fp32. */
#define GFN_F32 0
#define GFN_F16 1

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
 *   f1   (B,C,H,W)  contiguous; or, for a symmetric batch (model/network.py:213-222 concatenates
 *                   (A,B) against (B,A)), f1 = the B-image maps (B/2,C,H,W) used by directions
 *                   b < B/2 and f1_second = the A-image maps used by b >= B/2 -- no concatenated
 *                   copy is needed.  f1_second == NULL: plain batch.
 *   flow (B,2,G,G)  contiguous, or NULL
 *   out  (B,K,G,G)  batch stride out_bs floats (>= K*G*G), K = (2r+1)^2
 *
 * Fast path (LDS-tiled, shared bilinear fractions): C % 16 == 0, 1 <= r <= 7, !grid_based,
 * win_h == H, win_w == W, and `scratch` (device memory, gfn_local_corr_scratch_bytes(B, G) bytes,
 * 4-byte aligned; holds the list of tiles whose search windows do not fit the LDS stage and are
 * finished by a second, gather-based launch.  Its first 16 bytes are counters that must be ZERO on entry: zero
 * them once after allocation -- every successful call leaves them zero again (the second launch resets them, so
 * no memset precedes a call); after a call that returned an error, zero them again before reuse).  Anything else -- including scratch == NULL -- runs
 * the general per-tap kernel.  Flow values are unrestricted (out-of-image taps read zeros).
 */
int64_t gfn_local_corr_scratch_bytes(int B, int G);
int gfn_local_corr_fwd(const float *f0, int64_t f0_bs, const float *f1, const float *f1_second, const float *flow,
                       float *out, int64_t out_bs, int B, int C, int G, int H, int W, int r, int grid_based, int win_h,
                       int win_w, void *scratch, int64_t scratch_bytes, gfn_stream_t stream);

/* Variant selector for experiments/tests: 0 = auto (as above: the lean tile path of csrc/local_corr_lean.h for r <= 4; for r >= 5 on
 * 64-channel maps the matrix-core tile kernel of csrc/local_corr_mq.h -- split-bf16 products with fp32 accumulation, NOT bit-identical to
 * the fp32 FMA kernels: each product is exact to 2^-17 relative, so a value is within 2^-17 * sum_c |f0_c * f1_c| / sqrt(C) of the fp32
 * result (a few 1e-6, i.e. inside 1e-4 * max(1, |ref|), on unit-scale features; the bound follows the operands' magnitude, not the
 * result's; variant 4 is the opt-out, gfnet_amd.ops.LOCAL_CORR_FP32 in Python) -- and the round-1 tile kernel above for other channel counts), 1 = force the general per-tap kernel, 2 = the
 * round-1 fp32 tile kernel for every radius (the cross-check of the other paths), 4 = fp32 FMA arithmetic whatever the radius: the lean
 * tile path for r <= 4, the round-1 kernel for r >= 5 (bit-identical to 2); + 8: the plan of this call is already in scratch
 * (gfn_refiner_input_plan_fwd_dt; only with 0).  Same arguments otherwise.
 * Scratch header (ints): [0] tiles left to the second launch, [1..2] its queue counters, [3] last call's [0], [4] cells redone
 * per tap, [5] last call's [4], [6] tiles staged as two halves (sampled), [7] last call's [6]; [0..2], [4], [6] are zero
 * between calls.  [4]..[7] are informational and approximate: the r >= 3 kernels publish and reset them from inside the launch
 * while other workgroups may still be adding, so a handful of flagged cells can be attributed to the next call. */
int gfn_local_corr_fwd_ex(const float *f0, int64_t f0_bs, const float *f1, const float *f1_second, const float *flow,
                          float *out, int64_t out_bs, int B, int C, int G, int H, int W, int r, int grid_based, int win_h,
                          int win_w, int variant, void *scratch, int64_t scratch_bytes, gfn_stream_t stream);

/* The same with the feature map f1 (and f1_second) stored as f1_dtype = GFN_F32 or GFN_F16 (read directly, widened in
 * registers; f0, flow and out stay fp32). */
int gfn_local_corr_fwd_dt(const float *f0, int64_t f0_bs, const void *f1, const void *f1_second, int f1_dtype, const float *flow,
                          float *out, int64_t out_bs, int B, int C, int G, int H, int W, int r, int grid_based, int win_h,
                          int win_w, int variant, void *scratch, int64_t scratch_bytes, gfn_stream_t stream);

/* Backward of gfn_local_corr_fwd with respect to f0 -- the only gradient the reference lets through
 * (utils/local_correlation.py:54-60: sampling coordinates and feature1 are used under no_grad):
 *   grad_f0[b,c,i,j] = (sum_k grad_out[b,k,i,j] * bilinear(f1[b,c], tap k of cell (i,j))) / sqrt(C).
 * Same arguments as the forward; grad_out (B,K,G,G) with batch stride grad_out_bs, grad_f0 (B,C,G,G) with grad_f0_bs. */
int gfn_local_corr_bwd_f0(const float *grad_out, int64_t grad_out_bs, const float *f1, const float *f1_second, const float *flow,
                          float *grad_f0, int64_t grad_f0_bs, int B, int C, int G, int H, int W, int r, int grid_based, int win_h,
                          int win_w, gfn_stream_t stream);

/* F.avg_pool2d(x, 2, 2) between correlation levels (utils/local_correlation.py:71).
 * in (BC,H,W) -> out (BC,H/2,W/2). */
int gfn_avg_pool2(const float *in, float *out, int BC, int H, int W, gfn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Global correlation + soft-argmax -- model/network.py:415-428 (GFNet.corr_volume) and :430-440
 * (GFNet.pos_embed), called at :251-252.
 *   V[b,j,i]     = sum_c f0[b,c,i] * f1[b,c,j] / sqrt(C)        vol: (B,H1,W1,H0,W0)
 *   flow[b,:,i]  = sum_j softmax_j(V[b,j,i]) * (x_j, y_j)        flow: (B,2,H0,W0)
 * gfn_corr_softargmax_fwd is the fused form (the volume is never written); with symmetric != 0
 * f0/f1 hold B/2 images each and directions b >= B/2 swap their roles (the reference's
 * cat((A,B)) vs cat((B,A)) batch, model/network.py:213-222, without the copies);
 * gfn_corr_volume_fwd writes the volume (and the flow too when flow_or_null != NULL);
 * gfn_pos_embed_fwd is pos_embed on a caller-supplied volume.  f0 (B,C,H0,W0), f1 (B,C,H1,W1),
 * C <= 128.
 */
int gfn_corr_softargmax_fwd(const float *f0, const float *f1, float *flow, int B, int C, int H0, int W0, int H1, int W1,
                            int symmetric, gfn_stream_t stream);
/* The same with the feature maps stored as dtype = GFN_F32 or GFN_F16 (widened on load; products and sums fp32). */
int gfn_corr_softargmax_fwd_dt(const void *f0, const void *f1, int dtype, float *flow, int B, int C, int H0, int W0, int H1, int W1,
                               int symmetric, gfn_stream_t stream);
/* The same with a caller-owned workspace (round 6).  On 33..64-channel maps whose rows are 32..64 positions wide (GFNet's stride-16
 * features, model/network.py:251-252) the products run on the bf16 matrix core instruction with both operands split three ways
 * (x = h + m + l exactly; six of the nine piece products kept, the dropped ones <= 2^-23 of |a||b|: measured closer to a float64
 * evaluation than the fp32 fma chains of the calls above); with ws_bytes >= gfn_corr_softargmax_ws_bytes(B, C, H1, W1) the
 * B-positions' operand is split ONCE into ws (16-byte aligned device memory, contents undefined afterwards) instead of by every
 * wave that walks it.  ws == NULL or too small: same results, slower.  gfn_corr_softargmax_ws_bytes returns 0 for shapes that
 * take no workspace. */
int64_t gfn_corr_softargmax_ws_bytes(int B, int C, int H1, int W1);
int gfn_corr_softargmax_fwd_ws(const void *f0, const void *f1, int dtype, float *flow, int B, int C, int H0, int W0, int H1, int W1,
                               int symmetric, void *ws, int64_t ws_bytes, gfn_stream_t stream);
int gfn_corr_volume_fwd(const float *f0, const float *f1, float *vol, float *flow_or_null, int B, int C, int H0, int W0,
                        int H1, int W1, gfn_stream_t stream);
int gfn_pos_embed_fwd(const float *vol, float *flow, int B, int H0, int W0, int H1, int W1, gfn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * ConvRefiner.forward prefix -- model/network.py:533-555.  Writes, into the concat buffer
 * d (B, 2C+disp_dim[+K], G, G) with batch stride d_bs:
 *   d[:, 0:C]          = grid_sample(f0, cell centres)                    (:539-547)
 *   d[:, C:2C]         = grid_sample(f1, flow)                            (:537)
 *   d[:, 2C:2C+disp]   = disp_w @ (disp_scale * (flow - centres)) + disp_b (:548-549),
 *                        disp_scale = 40/32 * scale_factor, disp_w (disp_dim,2), disp_b (disp_dim)
 * The local-correlation slice d[:, 2C+disp:] is filled by gfn_local_corr_fwd with
 * f0 = d (batch stride d_bs) and out = d + (2C+disp)*G*G (batch stride d_bs).
 * symmetric bit 0 (1): f0/f1 hold B/2 images each; direction b < B/2 queries f0[b] against f1[b],
 * direction b >= B/2 queries f1[b-B/2] against f0[b-B/2].
 * symmetric bit 1 (2, GFN_RI_KEEP_GRID_FEATURE): d[:, 0:C] is left as it is -- grid_sample(f0, cell centres) depends on f0
 * and G only, so the second refiner iteration at a scale (num_itr = 2, model/network.py:257-268: same features, new flow)
 * passes the d of the first one and has only x_hat and the displacement embedding rewritten.
 */
#define GFN_RI_KEEP_GRID_FEATURE 2
int gfn_refiner_input_fwd(const float *f0, const float *f1, const float *flow, const float *disp_w, const float *disp_b,
                          float *d, int64_t d_bs, int B, int C, int Hs, int Ws, int G, int disp_dim, float disp_scale,
                          int symmetric, gfn_stream_t stream);
/* The same with the two feature maps stored as dtype = GFN_F32 or GFN_F16 (gathered directly, widened in registers; d is fp32). */
int gfn_refiner_input_fwd_dt(const void *f0, const void *f1, int dtype, const float *flow, const float *disp_w, const float *disp_b,
                             float *d, int64_t d_bs, int B, int C, int Hs, int Ws, int G, int disp_dim, float disp_scale,
                             int symmetric, gfn_stream_t stream);
/* gfn_refiner_input_fwd_dt that, in the same launch, also plans the tiles of the local correlation that follows it in
 * ConvRefiner.forward (model/network.py:537-555; both only read the flow): call gfn_local_corr_fwd_dt next with the same B, C,
 * G, Hs, Ws, r, flow, dtype and scratch and variant = 8 ("plan present"), which then skips its own plan launch.  Only for shapes
 * with gfn_local_corr_plans(...) != 0 (the lean tile path: 1 <= r <= 4, C in {16, 32, 64}, ...); scratch as for
 * gfn_local_corr_fwd (gfn_local_corr_scratch_bytes(B, G) bytes, counters zero). */
int gfn_local_corr_plans(int C, int H, int W, int G, int r, int f1_dtype);
int gfn_refiner_input_plan_fwd_dt(const void *f0, const void *f1, int dtype, const float *flow, const float *disp_w, const float *disp_b,
                                  float *d, int64_t d_bs, int B, int C, int Hs, int Ws, int G, int disp_dim, float disp_scale,
                                  int symmetric, int r, void *scratch, int64_t scratch_bytes, gfn_stream_t stream);

/* F.grid_sample(in, grid, mode='bilinear', padding_mode='zeros', align_corners=False):
 * in (B,C,H,W), grid (B,Ho,Wo,2) -> out (B,C,Ho,Wo) with batch stride out_bs. */
int gfn_grid_sample_fwd(const float *in, const float *grid, float *out, int64_t out_bs, int B, int C, int H, int W,
                        int Ho, int Wo, gfn_stream_t stream);

/* F.interpolate(x, size=(Ho,Wo), mode='bilinear', align_corners=False) -- model/network.py:238-249,
 * 271-281, 333-335.  in (BC,H,W) -> out (BC,Ho,Wo). */
int gfn_interp_bilinear_fwd(const float *in, float *out, int BC, int H, int W, int Ho, int Wo, gfn_stream_t stream);
/* The same for two tensors of one spatial size in a single launch -- the flow (B,2,H,W) and certainty (B,1,H,W) pair the
 * scale loop resizes together at model/network.py:238-249 and 271-281. */
int gfn_interp_bilinear_pair_fwd(const float *in_a, float *out_a, int BCa, const float *in_b, float *out_b, int BCb, int H, int W,
                                 int Ho, int Wo, gfn_stream_t stream);

/* Flow/certainty accumulation of one refiner iteration -- model/network.py:262-268, in place:
 *   disp = scale * (delta[:,0]/(4*W0), delta[:,1]/(4*H0)); eval mode (zero_small): components with
 *   |disp - disp_prev| / |disp_prev| < 1e-6 are zeroed (disp_prev = 1e-7 when first_iteration);
 *   flow += disp; certainty += delta[:,2]; disp_prev <- disp.
 * flow (B,2,G,G), certainty (B,1,G,G), delta (B,>=3,G,G) with batch stride delta_bs, disp_prev (B,2,G,G). */
int gfn_flow_update_fwd(float *flow, float *certainty, const float *delta, int64_t delta_bs, float *disp_prev, int B, int G,
                        int scale, int W0, int H0, int zero_small, int first_iteration, gfn_stream_t stream);
/* The same step out of place and with the refiner's two outputs as they come (network.py:259-268 keeps every iteration's
 * flow and certainty): flow_out = flow_in + disp(dflow), cert_out = cert_in + dcert; dflow (B,>=2,G,G) with batch stride
 * dflow_bs, dcert (B,>=1,G,G) with dcert_bs.  Outputs may alias the inputs.  disp_prev may be NULL when first_iteration is
 * set and no further iteration follows at this scale (the displacement is then not stored). */
int gfn_flow_update_out_fwd(const float *flow_in, const float *cert_in, float *flow_out, float *cert_out, const float *dflow,
                            int64_t dflow_bs, const float *dcert, int64_t dcert_bs, float *disp_prev, int B, int G, int scale,
                            int W0, int H0, int zero_small, int first_iteration, gfn_stream_t stream);

/* match() post-processing -- model/network.py:332-338 + 358-384.
 *   flow (nb,2,G,G), certainty (nb,1,G,G) finest-scale logits, nb = 2*B_images when symmetric;
 *   cert16_or_null (nb,1,Gc,Gc): scale-16 certainty for the attenuation term, or NULL;
 *   warp (B_images, G, Gw, 4), cert_out (B_images, G, Gw), Gw = 2G (symmetric) or G. */
int gfn_match_post_fwd(const float *flow, const float *certainty, const float *cert16_or_null, float *warp, float *cert_out,
                       int B_images, int G, int Gc, int symmetric, gfn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Gaussian KDE -- utils/kde.py:4-13 (called from GFNet.sample, model/network.py:408).
 *   out[bt,n] = sum_m exp(-|x[bt,n] - y[bt,m]|^2 / (2 std^2))
 * x (Bt,N,D) contiguous; y: Bt blocks of M rows, row stride y_row_stride floats (= down*D for
 * x[::down]), batch stride y_batch_stride floats; out (Bt,N).  scratch: optional device buffer of
 * gfn_kde_scratch_floats(...) floats (pre-scaled copies + split-M partial sums); with less the
 * call falls back to a slower single-pass kernel but stays correct.
 */
int gfn_kde_msplit(int Bt, int N, int M);
int64_t gfn_kde_scratch_floats(int Bt, int N, int M, int D);
int gfn_kde_density(const float *x, const float *y, float *out, int Bt, int N, int M, int D, int64_t y_row_stride,
                    int64_t y_batch_stride, double std, float *scratch, int64_t scratch_floats, gfn_stream_t stream);

/* Spatially culled KDE for 4-D matches (same sum as gfn_kde_density up to terms below 2^-32): the
 * caller orders x and y by gfn_kde_morton_keys (any stable sort of the int keys) and passes the
 * sorted arrays; out is in the order of the sorted x (or the original one, see perm).  scratch: gfn_kde_sorted_scratch_floats().
 * Blocks of 64 reference points farther than 6.7 std from a wave's 64 queries are skipped. */
int gfn_kde_morton_keys(const float *x, int *keys, int64_t n, gfn_stream_t stream);
/* The order itself: rows of x (Bt,N,4) stably sorted by their Morton key, one launch.  x_sorted (Bt,N,4),
 * perm (Bt,N): x_sorted[b][i] = x[b][perm[b][i]]; scratch: Bt*N ints. */
int gfn_kde_morton_sort(const float *x, float *x_sorted, int *perm, int *scratch, int Bt, int N, gfn_stream_t stream);
int64_t gfn_kde_sorted_scratch_floats(int Bt, int N, int M);
/* perm (Bt,N) from gfn_kde_morton_sort of x, or NULL: with it the densities are written in the caller's original order
 * (out[b][perm[b][i]] = density of x_sorted[b][i]); without it, in the sorted order. */
int gfn_kde_density_sorted(const float *x, const float *y, float *out, const int *perm, int Bt, int N, int M, double std,
                           int round_fp16, float *scratch, int64_t scratch_floats, gfn_stream_t stream);
/* round_fp16: coordinates rounded to fp16 before use, as GFNet.sample hands them to kde() (network.py:408, kde.py:6). */

/* GFNet.sample's elementwise steps (model/network.py:385-414):
 *   gfn_threshold_certainty: out = certainty > thresh ? 1 : certainty            (:391-393)
 *   gfn_balance_weights:     p = density < min_density ? floor_p : 1/(density+1)  (:409-410; 10, 1e-7)
 * (the two torch.multinomial draws stay with torch's generator, as in the reference). */
int gfn_threshold_certainty(const float *certainty, float *out, int64_t n, float thresh, gfn_stream_t stream);
int gfn_balance_weights(const float *density, float *p, int64_t n, float min_density, float floor_p, int round_fp16,
                        gfn_stream_t stream);  /* round_fp16: density rounded to fp16 first (kde(half=True) returns fp16, kde.py:13) */

/* Weighted sampling without replacement -- the two torch.multinomial(p, num_samples, replacement=False) draws of
 * GFNet.sample (model/network.py:400-402, 411-413).  Exponential race like torch's implementation (key = w / Exp(1),
 * the num_samples largest keys win) with a counter-based generator: same distribution, its own random stream, the same
 * result for the same seed.  weights (Bt,N) with row stride row_stride (non-negative; zero-weight entries are drawn only
 * when fewer than K positive ones exist), out (Bt,K) int64 indices in increasing order, scratch: Bt*(N + 2048) ints.
 * one_above: weights above it count as 1 -- the certainty threshold of model/network.py:391-393 applied on the fly
 * (+INFINITY: off). */
int gfn_sample_without_replacement(const float *weights, int64_t row_stride, int64_t *out, int *scratch, int Bt, int N, int K,
                                   uint64_t seed, float one_above, gfn_stream_t stream);
/* The gathers that follow a draw (model/network.py:403-404, 414): out_matches[b][i] = matches[b][idx[b][i]] (rows of four
 * floats, 16-byte aligned), out_certainty[b][i] = certainty[b][idx[b][i]], thresholded like above.  matches (Bt,N,4),
 * certainty (Bt,N), idx (Bt,K) int64. */
int gfn_gather_matches(const float *matches, const float *certainty, const int64_t *idx, float *out_matches, float *out_certainty,
                       int Bt, int N, int K, float one_above, gfn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Homography solve -- estimation.py:60-77 (cv2.findHomography(..., cv2.RANSAC, confidence=0.99999,
 * ransacReprojThreshold=3) in the reference; OpenCV's published pipeline restated, see
 * oracle/homography_oracle.c) and estimation.py:26-45 (convert_coordinates).
 *
 * gfn_convert_matches: matches (n,4) normalised warp rows -> pts (n,4) pixel (x,y,u,v), float32,
 *   (w-1)*(x+1)/2 per coordinate.
 * gfn_homography_ransac: pts (Bt,N,4) pixels -> H (Bt,9) row-major double (H[8] = 1),
 *   ninl (Bt) inlier count of the chosen hypothesis, best_t (Bt) its index or -1,
 *   mask (Bt,N) bytes or NULL.  Failure (fewer than 4 inliers / degenerate) gives diag(0,0,1),
 *   the reference's convention (estimation.py:74-76).  stage: 0 = RANSAC -> DLT on inliers ->
 *   lm_iters LM steps; 1 = RANSAC only; 2 = RANSAC + DLT.  scratch: gfn_homography_scratch_bytes().
 * gfn_homography_dlt: one-shot weighted normalised DLT ("grid-DLT"): weight (Bt,N) float or NULL;
 *   ok (Bt) = 1 on success.
 */
int gfn_convert_matches(const float *matches, float *pts, int64_t n, float wA, float hA, float wB, float hB,
                        gfn_stream_t stream);
int64_t gfn_homography_scratch_bytes(int Bt, int iters);
int gfn_homography_ransac(const float *pts, int Bt, int N, double thresh, int iters, uint64_t seed, int lm_iters, int stage,
                          double *H, int *ninl, int *best_t, unsigned char *mask, void *scratch, int64_t scratch_bytes,
                          gfn_stream_t stream);
/* The same with OpenCV's termination rule (cv::RANSACPointSetRegistrator::run): 0 < confidence < 1 -- whenever a hypothesis
 * beats the best inlier count so far the iteration bound becomes log(1 - confidence) / log(1 - w^4), w = its inlier ratio
 * (estimation.py:66-72 passes 0.99999); confidence <= 0 scores all `iters` hypotheses like gfn_homography_ransac.  In both
 * modes a 4-point subset with collinear points or inconsistent orientation is re-drawn (checkSubset).  iters_used (Bt) or
 * NULL: the bound each pair stopped at. */
int gfn_homography_ransac_ex(const float *pts, int Bt, int N, double thresh, int iters, double confidence, uint64_t seed, int lm_iters,
                             int stage, double *H, int *ninl, int *best_t, unsigned char *mask, int *iters_used, void *scratch,
                             int64_t scratch_bytes, gfn_stream_t stream);
int gfn_homography_dlt(const float *pts, const float *weight, int Bt, int N, double *H, int *ok, gfn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Image resize + normalise in front of the backbone -- GFNet.match's
 * get_tuple_transform_ops(resize, mode, normalize=True), model/network.py:293-346 with utils/utils.py:18-27,
 * 87-116: torchvision Resize on a float tensor with antialias=None (= F.interpolate(mode,
 * align_corners=False), no antialiasing) followed by Normalize on the first three channels.
 *   in (B, >=3, H, W) floats in [0,1], batch stride in_bs floats; out (B,3,Ho,Wo) = (resize(in[:, :3]) - mean) / std;
 *   mode 0 = bilinear (the reference's mode=2: path inputs and the upsample pass), 1 = bicubic (PIL / tensor
 *   inputs); mean3 / std3: three host floats each (ImageNet statistics in the reference).
 */
int gfn_resize_normalize_fwd(const float *in, int64_t in_bs, float *out, int B, int H, int W, int Ho, int Wo, int mode,
                             const float *mean3, const float *std3, gfn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Refiner conv stack -- ConvRefiner.create_block / forward, model/network.py:471-487
 * and :560-563: nine blocks of depthwise 5x5 conv -> BatchNorm2d(eval) -> ReLU -> 1x1 conv, then a
 * final 1x1 conv to 3 channels.  Depthwise, norm and accumulation always fp32; the 1x1 products fp32
 * (default) or fp16.
 *
 * gfn_conv_block_pack: lays one block's parameters out for the kernels (device to device):
 *   dw_w (C,25) depthwise taps, dw_b (C) or NULL, bn_alpha/bn_beta (C) the eval-mode BatchNorm as
 *   y = x*alpha + beta (alpha = weight/sqrt(running_var+eps), beta = bias - running_mean*alpha),
 *   pw_w (M,C) and pw_b (M) the 1x1 conv; packed: gfn_conv_block_packed_floats(C, M) floats.
 * gfn_conv_block_fwd: y = pw(relu(bn(dw(x)))) for x (B,C,G,G) -> y (B,M,G,G), zero padding 2; y must
 *   not alias x.  variant bit 0 (1): two-pass form (depthwise kernel -> t_scratch (B*C*G*G floats) ->
 *   GEMM kernel) instead of the fused kernel; also taken when G % 4 != 0; bit-identical results.
 *   variant bit 1 (2): the 1x1 conv takes fp16 operands (W and relu output rounded to nearest fp16,
 *   v_mfma_f32_32x32x16_f16, fp32 accumulation) -- the reference's autocast numerics class for its
 *   amp=True refiners (model/network.py:560-562) -- instead of fp32 throughout.  t_scratch may be NULL
 *   when the fused kernel applies.
 * gfn_pointwise_conv_fwd: y[b] = W . t[b] + bias for a few output channels (M <= 16; the final C -> 3
 *   conv, network.py:505,563): w (M,K), t (B,K,N), y (B,M,N).
 */
int64_t gfn_conv_block_packed_floats(int C, int M);
int gfn_conv_block_pack(const float *dw_w, const float *dw_b, const float *bn_alpha, const float *bn_beta, const float *pw_w,
                        const float *pw_b, float *packed, int C, int M, gfn_stream_t stream);
int gfn_conv_block_fwd(const float *x, const float *packed, float *y, float *t_scratch, int B, int C, int M, int G, int variant,
                       gfn_stream_t stream);
int gfn_pointwise_conv_fwd(const float *w, const float *bias, const float *t, float *y, int B, int M, int K, int N,
                           gfn_stream_t stream);

/* gfn_conv_block_half_fwd: the same block with fp16 maps in HBM -- the reference's amp=True refiners, where every map
 *   between two blocks is a float16 tensor (torch.autocast around block1 + hidden_blocks, model/network.py:560-562).
 *   Arithmetic = the autocast class: the depthwise 5x5 runs on the matrix core too (v_mfma_f32_16x16x32_f16: the input halo --
 *   also a first block's fp32 concat -- and the folded taps rounded to fp16, fp32 accumulation), BatchNorm + ReLU fp32, ReLU output
 *   and 1x1 weights fp16, fp32 accumulation; builds with -DGFN_CONV_VALU_DW keep the depthwise in fp32 as gfn_conv_block_fwd
 *   variant 2 does.  A map of
 *   dtype GFN_F16 is (B, ceil(C/2), G, G) of half2: channels 2p and 2p+1 of a cell side by side, the odd channel past C zero
 *   (the kernel writes it so).  x_dtype / y_dtype: GFN_F32 (B, C, G, G) floats or GFN_F16; at least one of them GFN_F16 (a
 *   stack's first block reads the fp32 concat, its last one writes fp32).  G must be a multiple of 4. */
int gfn_conv_block_half_fwd(const void *x, int x_dtype, const float *packed, void *y, int y_dtype, int B, int C, int M, int G,
                            gfn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GFNET_HIP_H */
