// grid_ops.hip -- the small gather / elementwise kernels around the correlation on the hot path
// (gfx950).  All are HBM/L2-bound one-pass kernels with coalesced stores along the grid row.
//
//   gfn_refiner_input_fwd  ConvRefiner.forward prefix, model/network.py:533-555: the two
//                          grid_samples, the displacement embedding (1x1 conv of 2 channels) --
//                          written straight into the channel slices of the concat buffer `d`
//                          (the local-correlation kernel fills the last slice), so the
//                          reference's torch.cat copy of up to 417 channels disappears.
//   gfn_grid_sample_fwd    F.grid_sample(bilinear, zeros, align_corners=False)
//   gfn_interp_bilinear_fwd F.interpolate(mode='bilinear', align_corners=False), network.py:238-249,271-281
//   gfn_flow_update_fwd    displacement scaling / eval-time zeroing / accumulation, network.py:262-268
//   gfn_match_post_fwd     certainty attenuation, sigmoid, out-of-range masking, clamp, warp
//                          assembly, network.py:332-338 + 358-384
#include "common.h"
#include "refiner_input.h"

namespace {

using namespace gfn_ri;

template <typename FT, bool KEEP>
__global__ __launch_bounds__(256) void refiner_input_kernel(RiArgs q, unsigned q_blocks, int banded) {
    if (banded) {  // 1-D grid, XCD-banded order (refiner_input.h)
        int b;
        unsigned x;
        if (ri_banded(blockIdx.x, q_blocks, q.Bh, b, x)) refiner_input_cell<FT, KEEP>(q, b, x * 256u + threadIdx.x);
        return;
    }
    refiner_input_cell<FT, KEEP>(q, ri_direction(q.B, q.Bh, blockIdx.y), blockIdx.x * 256u + threadIdx.x);
}

__global__ __launch_bounds__(256) void grid_sample_kernel(const float *__restrict__ in, const float *__restrict__ grid,
                                                          float *__restrict__ out, long out_bs, int B, int C, int H, int W,
                                                          int Ho, int Wo) {
    const long total = (long)B * C * Ho * Wo;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int j = (int)(idx % Wo);
        long t = idx / Wo;
        const int i = (int)(t % Ho);
        t /= Ho;
        const int c = (int)(t % C);
        const int b = (int)(t / C);
        const float *g = grid + (((size_t)b * Ho + i) * Wo + j) * 2;
        const Bilin s = bilin_setup(g[0], g[1], W, H);
        out[(size_t)b * out_bs + ((size_t)c * Ho + i) * Wo + j] = bilin_fetch(in + ((size_t)b * C + c) * H * W, W, s);
    }
}

// ATen upsample_bilinear2d, align_corners=False: src = max(0, (dst+0.5)*in/out - 0.5)
__device__ __forceinline__ float interp_at(const float *pl, int H, int W, int Ho, int Wo, int y, int x) {
    const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;
    float fy = ((float)y + 0.5f) * sy - 0.5f, fx = ((float)x + 0.5f) * sx - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    return hy * (hx * pl[(size_t)y0 * W + x0] + lx * pl[(size_t)y0 * W + x1]) +
           ly * (hx * pl[(size_t)y1 * W + x0] + lx * pl[(size_t)y1 * W + x1]);
}

// ---- image resize + normalise ----------------------------------------------------
// ATen upsample_bicubic2d, align_corners=False: src = (dst+0.5)*in/out - 0.5 (not clamped), taps floor(src)-1 .. +2 with
// clamped indices, cubic convolution coefficients with A = -0.75, rows first then columns.
__device__ __forceinline__ float cubic1(float x) { return ((-0.75f + 2.f) * x - (-0.75f + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x) { return ((-0.75f * x - 5.f * -0.75f) * x + 8.f * -0.75f) * x - 4.f * -0.75f; }
__device__ __forceinline__ void cubic_coeffs(float t, float c[4]) {
    c[0] = cubic2(t + 1.f);
    c[1] = cubic1(t);
    c[2] = cubic1(1.f - t);
    c[3] = cubic2(2.f - t);
}

struct NormParams {
    float mean[3], std[3];
};

// in (B, >=3, H, W) with batch stride in_bs; out (B, 3, Ho, Wo) = (resize(in[:, :3]) - mean) / std.  One thread per output
// pixel (the three channels share indices and coefficients).  mode 0 bilinear, 1 bicubic.
__global__ __launch_bounds__(256) void resize_normalize_kernel(const float *__restrict__ in, long in_bs, float *__restrict__ out,
                                                               int B, int H, int W, int Ho, int Wo, int mode, NormParams np) {
    const long total = (long)B * Ho * Wo;
    const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;
    const bool same = H == Ho && W == Wo;  // transforms.Resize returns the image untouched
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % Wo);
        const long t = idx / Wo;
        const int y = (int)(t % Ho), b = (int)(t / Ho);
        const float *src = in + (size_t)b * in_bs;
        float v[3];
        if (same) {
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = src[((size_t)c * H + y) * W + x];
        } else if (mode == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = interp_at(src + (size_t)c * H * W, H, W, Ho, Wo, y, x);
        } else {
            const float fy = ((float)y + 0.5f) * sy - 0.5f, fx = ((float)x + 0.5f) * sx - 0.5f;
            const float flx = floorf(fx), fly = floorf(fy);
            const int ix = (int)flx, iy = (int)fly;
            float cx[4], cy[4];
            cubic_coeffs(fx - flx, cx);
            cubic_coeffs(fy - fly, cy);
            int xs[4], ys[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xs[i] = min(max(ix - 1 + i, 0), W - 1);
                ys[i] = min(max(iy - 1 + i, 0), H - 1);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float *pl = src + (size_t)c * H * W;
                float rows[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float *r = pl + (size_t)ys[i] * W;
                    rows[i] = r[xs[0]] * cx[0] + r[xs[1]] * cx[1] + r[xs[2]] * cx[2] + r[xs[3]] * cx[3];
                }
                v[c] = rows[0] * cy[0] + rows[1] * cy[1] + rows[2] * cy[2] + rows[3] * cy[3];
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) out[(((size_t)b * 3 + c) * Ho + y) * Wo + x] = (v[c] - np.mean[c]) / np.std[c];
    }
}

// Four consecutive outputs of one row per thread (one 16-byte store when the row pitch allows), 32-bit index arithmetic,
// planes on blockIdx.y: the one-output-per-thread form with 64-bit div/mod ran at 1 TB/s.  Same per-pixel arithmetic
// (interp_at) as before.
__device__ __forceinline__ void interp_quad(const float *__restrict__ pl, float *__restrict__ oplane, int H, int W, int Ho, int Wo,
                                            unsigned q, int Wq) {
    const int y = (int)(q / (unsigned)Wq), x0 = ((int)q - y * Wq) * 4;
    float *o = oplane + (size_t)y * Wo + x0;
    if (x0 + 3 < Wo && (Wo & 3) == 0 && (((uintptr_t)oplane & 15) == 0)) {
        float4 v;
        v.x = interp_at(pl, H, W, Ho, Wo, y, x0);
        v.y = interp_at(pl, H, W, Ho, Wo, y, x0 + 1);
        v.z = interp_at(pl, H, W, Ho, Wo, y, x0 + 2);
        v.w = interp_at(pl, H, W, Ho, Wo, y, x0 + 3);
        *reinterpret_cast<float4 *>(o) = v;
    } else {
        for (int k = 0; k < 4 && x0 + k < Wo; ++k) o[k] = interp_at(pl, H, W, Ho, Wo, y, x0 + k);
    }
}

__global__ __launch_bounds__(256) void interp_bilinear_kernel(const float *__restrict__ in, float *__restrict__ out, int BC,
                                                              int H, int W, int Ho, int Wo) {
    const int Wq = (Wo + 3) >> 2;
    const unsigned q = blockIdx.x * 256u + threadIdx.x;
    if (q >= (unsigned)(Wq * Ho)) return;
    for (int pl = blockIdx.y; pl < BC; pl += gridDim.y)
        interp_quad(in + (size_t)pl * H * W, out + (size_t)pl * Ho * Wo, H, W, Ho, Wo, q, Wq);
}

// two tensors of the same spatial size in one launch (flow + certainty between scales: half the launches of the loop)
__global__ __launch_bounds__(256) void interp_bilinear_pair_kernel(const float *__restrict__ in_a, float *__restrict__ out_a, int BCa,
                                                                   const float *__restrict__ in_b, float *__restrict__ out_b, int BCb,
                                                                   int H, int W, int Ho, int Wo) {
    const int Wq = (Wo + 3) >> 2;
    const unsigned q = blockIdx.x * 256u + threadIdx.x;
    if (q >= (unsigned)(Wq * Ho)) return;
    for (int pl = blockIdx.y; pl < BCa + BCb; pl += gridDim.y) {
        const bool second = pl >= BCa;  // block-uniform
        const float *in = second ? in_b + (size_t)(pl - BCa) * H * W : in_a + (size_t)pl * H * W;
        float *out = second ? out_b + (size_t)(pl - BCa) * Ho * Wo : out_a + (size_t)pl * Ho * Wo;
        interp_quad(in, out, H, W, Ho, Wo, q, Wq);
    }
}

// dflow: (B, >=2, G, G) displacement increment with batch stride dflow_bs, dcert: (B, >=1, G, G) certainty increment with
// dcert_bs (the refiner's two outputs; one (B,3,G,G) tensor or two).  flow_in/cert_in -> flow_out/cert_out (may alias).
// flow_in / cert_in may be the same buffers as flow_out / cert_out (gfn_flow_update_fwd updates in place): no __restrict__ on them
__global__ __launch_bounds__(256) void flow_update_kernel(const float *flow_in, const float *cert_in,
                                                          float *flow_out, float *cert_out, const float *__restrict__ dflow,
                                                          long dflow_bs, const float *__restrict__ dcert, long dcert_bs,
                                                          float *__restrict__ disp_prev, int B, int G, float scale, float div_x,
                                                          float div_y, int zero_small, int first) {
    const long GG = (long)G * G, total = (long)B * GG;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int b = (int)(idx / GG);
        const long r = idx - (long)b * GG;
        const float *dl = dflow + (size_t)b * dflow_bs + r;
        float dx = scale * (dl[0] / div_x), dy = scale * (dl[GG] / div_y);  // network.py:262-263
        float *pp = disp_prev ? disp_prev + (size_t)b * 2 * GG + r : nullptr;  // NULL: single-iteration scale, nothing to carry
        if (zero_small) {  // network.py:256,264-265
            const float px = first ? 1e-7f : pp[0], py = first ? 1e-7f : pp[GG];
            if (fabsf(dx - px) / fabsf(px) < 1e-6f) dx = 0.f;
            if (fabsf(dy - py) / fabsf(py) < 1e-6f) dy = 0.f;
        }
        if (pp) {
            pp[0] = dx;
            pp[GG] = dy;
        }
        const size_t o = (size_t)b * 2 * GG + r;
        flow_out[o] = flow_in[o] + dx;
        flow_out[o + GG] = flow_in[o + GG] + dy;
        cert_out[idx] = cert_in[idx] + dcert[(size_t)b * dcert_bs + r];
    }
}

__global__ __launch_bounds__(256) void match_post_kernel(const float *__restrict__ flow, const float *__restrict__ cert,
                                                         const float *__restrict__ cert16, float *__restrict__ warp,
                                                         float *__restrict__ cert_out, int Bimg, int G, int Gc,
                                                         int symmetric) {
    const int Gw = symmetric ? 2 * G : G;
    const long total = (long)Bimg * G * Gw;
    const float lo = (float)(-1 + 1.0 / G), hi = (float)(1 - 1.0 / G);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int jw = (int)(idx % Gw);
        long t = idx / Gw;
        const int i = (int)(t % G);
        const int b = (int)(t / G);
        const bool second = jw >= G;  // the B->A half of a symmetric warp
        const int j = second ? jw - G : jw;
        const int fb = second ? Bimg + b : b;
        const size_t cell = (size_t)i * G + j, GG = (size_t)G * G;
        float fx = flow[((size_t)fb * 2 + 0) * GG + cell], fy = flow[((size_t)fb * 2 + 1) * GG + cell];
        float c = cert[(size_t)fb * GG + cell];
        if (cert16) {  // network.py:332-338
            const float low = interp_at(cert16 + (size_t)fb * Gc * Gc, Gc, Gc, G, G, i, j);
            c = c - 0.5f * low * (low < 0.f ? 1.f : 0.f);
        }
        c = 1.f / (1.f + expf(-c));                          // :361
        if (fabsf(fx) > 1.f || fabsf(fy) > 1.f) c = 0.f;     // :368-370
        fx = fminf(fmaxf(fx, -1.f), 1.f);                    // :371
        fy = fminf(fmaxf(fy, -1.f), 1.f);
        const float gx = gfn::linspace_at(lo, hi, G, j), gy = gfn::linspace_at(lo, hi, G, i);  // :362-367
        const float4 w = second ? make_float4(fx, fy, gx, gy) : make_float4(gx, gy, fx, fy);   // :373-378
        reinterpret_cast<float4 *>(warp)[idx] = w;
        cert_out[idx] = c;
    }
}

inline unsigned grid_for(long total, int cap = 16384) {
    long g = (total + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

GFN_EXPORT int gfn_refiner_input_fwd(const float *f0, const float *f1, const float *flow, const float *disp_w,
                                     const float *disp_b, float *d, int64_t d_bs, int B, int C, int Hs, int Ws, int G,
                                     int disp_dim, float disp_scale, int symmetric, gfn_stream_t stream) {
    return gfn_refiner_input_fwd_dt(f0, f1, GFN_F32, flow, disp_w, disp_b, d, d_bs, B, C, Hs, Ws, G, disp_dim, disp_scale, symmetric, stream);
}

GFN_EXPORT int gfn_refiner_input_fwd_dt(const void *f0, const void *f1, int dtype, const float *flow, const float *disp_w,
                                        const float *disp_b, float *d, int64_t d_bs, int B, int C, int Hs, int Ws, int G,
                                        int disp_dim, float disp_scale, int symmetric, gfn_stream_t stream) {
    if (dtype != GFN_F32 && dtype != GFN_F16) return gfn::fail(GFN_ERR_INVALID_ARG, "refiner_input: feature dtype must be GFN_F32 or GFN_F16");
    if (!f0 || !f1 || !flow || !d || (disp_dim > 0 && (!disp_w || !disp_b)))
        return gfn::fail(GFN_ERR_INVALID_ARG, "refiner_input: null pointer");
    if (B < 0 || C <= 0 || Hs <= 0 || Ws <= 0 || G <= 0 || disp_dim < 0 || d_bs < (int64_t)(2 * C + disp_dim) * G * G ||
        ((symmetric & 1) && (B & 1)) || (symmetric & ~3) || (long)C * Hs * Ws >= (1L << 31))
        return gfn::fail(GFN_ERR_INVALID_ARG, "refiner_input: bad size");
    if (B == 0) return GFN_OK;
    if (B > 65535 || (long)G * G >= (1L << 31)) return gfn::fail(GFN_ERR_INVALID_ARG, "refiner_input: batch > 65535 or grid too large");
    const unsigned q_blocks = (unsigned)(((long)G * G + 255) / 256);
    RiArgs q;
    q.fa = f0; q.fb = f1; q.flow = flow; q.dw = disp_w; q.db = disp_b; q.d = d; q.d_bs = (long)d_bs;
    q.B = B; q.Bh = (symmetric & 1) ? B / 2 : B; q.C = C; q.Hs = Hs; q.Ws = Ws; q.G = G; q.Dd = disp_dim; q.disp_scale = disp_scale;
    const bool keep = (symmetric & 2) != 0;  // the grid_feature planes are already in d (GFN_RI_KEEP_GRID_FEATURE)
    const int banded = gfn_ri::ri_bands(q.B, q.Bh, q_blocks) ? 1 : 0;
    const dim3 grid = banded ? dim3(gfn_ri::ri_banded_blocks(B, q_blocks)) : dim3(q_blocks, (unsigned)B);
    if (dtype == GFN_F16) {
        if (keep) hipLaunchKernelGGL((refiner_input_kernel<_Float16, true>), grid, dim3(256), 0, (hipStream_t)stream, q, q_blocks, banded);
        else hipLaunchKernelGGL((refiner_input_kernel<_Float16, false>), grid, dim3(256), 0, (hipStream_t)stream, q, q_blocks, banded);
    } else {
        if (keep) hipLaunchKernelGGL((refiner_input_kernel<float, true>), grid, dim3(256), 0, (hipStream_t)stream, q, q_blocks, banded);
        else hipLaunchKernelGGL((refiner_input_kernel<float, false>), grid, dim3(256), 0, (hipStream_t)stream, q, q_blocks, banded);
    }
    return gfn::check_launch("refiner_input_kernel");
}

GFN_EXPORT int gfn_grid_sample_fwd(const float *in, const float *grid, float *out, int64_t out_bs, int B, int C, int H,
                                   int W, int Ho, int Wo, gfn_stream_t stream) {
    if (!in || !grid || !out || B < 0 || C <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || out_bs < (int64_t)C * Ho * Wo)
        return gfn::fail(GFN_ERR_INVALID_ARG, "grid_sample: bad argument");
    if (B == 0) return GFN_OK;
    hipLaunchKernelGGL(grid_sample_kernel, dim3(grid_for((long)B * C * Ho * Wo)), dim3(256), 0, (hipStream_t)stream, in,
                       grid, out, (long)out_bs, B, C, H, W, Ho, Wo);
    return gfn::check_launch("grid_sample_kernel");
}

GFN_EXPORT int gfn_interp_bilinear_fwd(const float *in, float *out, int BC, int H, int W, int Ho, int Wo,
                                       gfn_stream_t stream) {
    if (!in || !out || BC < 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0)
        return gfn::fail(GFN_ERR_INVALID_ARG, "interp_bilinear: bad argument");
    if (BC == 0) return GFN_OK;
    if ((long)Ho * ((Wo + 3) / 4) >= (1L << 31)) return gfn::fail(GFN_ERR_INVALID_ARG, "interp_bilinear: output too large");
    const unsigned gx = (unsigned)(((long)Ho * ((Wo + 3) / 4) + 255) / 256);
    hipLaunchKernelGGL(interp_bilinear_kernel, dim3(gx, (unsigned)(BC < 65535 ? BC : 65535)), dim3(256), 0, (hipStream_t)stream, in,
                       out, BC, H, W, Ho, Wo);
    return gfn::check_launch("interp_bilinear_kernel");
}

GFN_EXPORT int gfn_interp_bilinear_pair_fwd(const float *in_a, float *out_a, int BCa, const float *in_b, float *out_b, int BCb, int H,
                                            int W, int Ho, int Wo, gfn_stream_t stream) {
    if (!in_a || !out_a || !in_b || !out_b || BCa < 0 || BCb < 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0)
        return gfn::fail(GFN_ERR_INVALID_ARG, "interp_bilinear_pair: bad argument");
    if (BCa + BCb == 0) return GFN_OK;
    if ((long)Ho * ((Wo + 3) / 4) >= (1L << 31)) return gfn::fail(GFN_ERR_INVALID_ARG, "interp_bilinear_pair: output too large");
    const unsigned gx = (unsigned)(((long)Ho * ((Wo + 3) / 4) + 255) / 256);
    const long planes = (long)BCa + BCb;
    hipLaunchKernelGGL(interp_bilinear_pair_kernel, dim3(gx, (unsigned)(planes < 65535 ? planes : 65535)), dim3(256), 0,
                       (hipStream_t)stream, in_a, out_a, BCa, in_b, out_b, BCb, H, W, Ho, Wo);
    return gfn::check_launch("interp_bilinear_pair_kernel");
}

GFN_EXPORT int gfn_resize_normalize_fwd(const float *in, int64_t in_bs, float *out, int B, int H, int W, int Ho, int Wo, int mode,
                                        const float *mean3, const float *std3, gfn_stream_t stream) {
    if (!in || !out || !mean3 || !std3 || B < 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || in_bs < 3L * H * W)
        return gfn::fail(GFN_ERR_INVALID_ARG, "resize_normalize: bad argument");
    if (mode != 0 && mode != 1) return gfn::fail(GFN_ERR_INVALID_ARG, "resize_normalize: mode must be 0 (bilinear) or 1 (bicubic)");
    NormParams np;
    for (int c = 0; c < 3; ++c) {
        if (!(std3[c] != 0.f)) return gfn::fail(GFN_ERR_INVALID_ARG, "resize_normalize: zero std");
        np.mean[c] = mean3[c];
        np.std[c] = std3[c];
    }
    if (B == 0) return GFN_OK;
    hipLaunchKernelGGL(resize_normalize_kernel, dim3(grid_for((long)B * Ho * Wo)), dim3(256), 0, (hipStream_t)stream, in, (long)in_bs,
                       out, B, H, W, Ho, Wo, mode, np);
    return gfn::check_launch("resize_normalize_kernel");
}

GFN_EXPORT int gfn_flow_update_fwd(float *flow, float *certainty, const float *delta, int64_t delta_bs, float *disp_prev,
                                   int B, int G, int scale, int W0, int H0, int zero_small, int first_iteration,
                                   gfn_stream_t stream) {
    if (!flow || !certainty || !delta || !disp_prev || B < 0 || G <= 0 || W0 <= 0 || H0 <= 0 || delta_bs < 3L * G * G)
        return gfn::fail(GFN_ERR_INVALID_ARG, "flow_update: bad argument");
    if (B == 0) return GFN_OK;
    hipLaunchKernelGGL(flow_update_kernel, dim3(grid_for((long)B * G * G)), dim3(256), 0, (hipStream_t)stream, flow, certainty, flow,
                       certainty, delta, (long)delta_bs, delta + 2L * G * G, (long)delta_bs, disp_prev, B, G, (float)scale,
                       (float)(4 * W0), (float)(4 * H0), zero_small, first_iteration);
    return gfn::check_launch("flow_update_kernel");
}

GFN_EXPORT int gfn_flow_update_out_fwd(const float *flow_in, const float *cert_in, float *flow_out, float *cert_out,
                                       const float *dflow, int64_t dflow_bs, const float *dcert, int64_t dcert_bs, float *disp_prev,
                                       int B, int G, int scale, int W0, int H0, int zero_small, int first_iteration,
                                       gfn_stream_t stream) {
    if (!flow_in || !cert_in || !flow_out || !cert_out || !dflow || !dcert || B < 0 || G <= 0 || W0 <= 0 || H0 <= 0 ||
        dflow_bs < 2L * G * G || dcert_bs < (long)G * G || (!disp_prev && !first_iteration))
        return gfn::fail(GFN_ERR_INVALID_ARG, "flow_update_out: bad argument");
    if (B == 0) return GFN_OK;
    hipLaunchKernelGGL(flow_update_kernel, dim3(grid_for((long)B * G * G)), dim3(256), 0, (hipStream_t)stream, flow_in, cert_in, flow_out,
                       cert_out, dflow, (long)dflow_bs, dcert, (long)dcert_bs, disp_prev, B, G, (float)scale, (float)(4 * W0),
                       (float)(4 * H0), zero_small, first_iteration);
    return gfn::check_launch("flow_update_kernel");
}

GFN_EXPORT int gfn_match_post_fwd(const float *flow, const float *certainty, const float *cert16_or_null, float *warp,
                                  float *cert_out, int B_images, int G, int Gc, int symmetric, gfn_stream_t stream) {
    if (!flow || !certainty || !warp || !cert_out || B_images < 0 || G <= 0 || (cert16_or_null && Gc <= 0))
        return gfn::fail(GFN_ERR_INVALID_ARG, "match_post: bad argument");
    if (B_images == 0) return GFN_OK;
    const long total = (long)B_images * G * (symmetric ? 2 * G : G);
    hipLaunchKernelGGL(match_post_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, flow, certainty,
                       cert16_or_null, warp, cert_out, B_images, G, Gc, symmetric);
    return gfn::check_launch("match_post_kernel");
}
