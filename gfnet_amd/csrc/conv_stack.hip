// conv_stack.hip -- the refiners' conv stack on gfx950 ("next" after the hot path).
//
// Reference: ConvRefiner.create_block / forward, model/network.py:471-487 and :560-563 -- per block
//   Conv2d(C, C, 5x5, padding 2, groups=C)  ->  BatchNorm2d (eval)  ->  ReLU  ->  Conv2d(C, C, 1x1)
// nine of them (block1 + 8 hidden blocks) and a final Conv2d(C, 3, 1x1), on (2c+disp+K)-channel grid
// maps: C = 417/361/177/73/24 at scales 16/8/4/2/1.  On MI355X the PyTorch/MIOpen stack takes
// 147 ms (fp16 autocast) / 180 ms (fp32) per 32-pair step -- 15x the whole correlation/sampling/solve
// path -- mostly in the small-C, large-grid scales where the depthwise convs are launch/latency bound.
//
// One conv block = one kernel (dwpw_fused_kernel): a workgroup owns a 128-cell tile of one map and a
// slab of output channels.  Per tile of 16 input channels it stages the cells' halo (zero padded) in
// LDS, computes t = relu((dw5x5(x) + conv_bias) * alpha + beta) on the VALU (alpha/beta = eval-mode
// BatchNorm folded by the packer) straight into the B-operand tile of the 1x1 conv, and runs
// y += W[:, k-tile] . t on the fp32 matrix core (v_mfma_f32_32x32x2_f32: exact fp32 products and
// accumulation, the numerics class of the reference's fp32 CPU path).  The intermediate t never
// leaves the CU: HBM traffic is one read of x and one write of y per block.  The next channel tile's
// halo and weight tile are prefetched into registers under the current tile's arithmetic.
//
// A two-pass variant (dw5x5 kernel -> t in HBM -> GEMM kernel) computes bit-identical results; it
// serves grids whose side is not a multiple of 4 and is the ablation/parity partner of the fused one.
#include "conv_block_fused.h"

namespace {


__global__ __launch_bounds__(256) void pack_block_kernel(const float *__restrict__ dw_w, const float *__restrict__ dw_b,
                                                         const float *__restrict__ alpha, const float *__restrict__ beta,
                                                         const float *__restrict__ pw_w, const float *__restrict__ pw_b,
                                                         float *__restrict__ packed, int C, int M) {
    const PackDims d(C, M);
    const size_t total = d.total();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (i < d.wt_off()) {
            const int pair = (int)(i / kCP2), j = (int)(i % kCP2);
            const int t = j >> 1, k = 2 * pair + (j & 1);
            if (k < C) {
                if (t < 25) v = dw_w[(size_t)k * 25 + t];
                else if (t == 25) v = dw_b ? dw_b[k] : 0.f;
                else if (t == 26) v = alpha[k];
                else if (t == 27) v = beta[k];
            }
        } else if (i < d.bias_off()) {
            const size_t e = i - d.wt_off();  // invert wt_index
            const int j = (int)(e & 3), m = (int)((e >> 2) % d.Mp), g = (int)((e >> 2) / d.Mp);
            const int k = (g >> 2) * 16 + ((g >> 1) & 1) * 8 + 2 * j + (g & 1);
            if (k < C && m < M) v = pw_w[(size_t)m * C + k];
        } else if (i < d.wt16_off()) {
            const int m = (int)(i - d.bias_off());
            if (m < M) v = pw_b[m];
        } else if (i >= d.mm_off()) {  // matrix-core depthwise parameters of a channel pair (PackDims::mm_off)
            const int pair = (int)((i - d.mm_off()) / kMM2), j = (int)((i - d.mm_off()) % kMM2);
            if (j < 80) {
                const int ch = j / 40, dy = (j % 40) >> 3, dx = j & 7, k = 2 * pair + ch;
                _Float16 h[2] = {(_Float16)0.f, (_Float16)0.f};
                if (dx < 5 && k < C) h[ch] = (_Float16)dw_w[(size_t)k * 25 + dy * 5 + dx];
                __builtin_memcpy(&v, h, 4);
            } else if (j < 86) {
                const int k = 2 * pair + (j & 1), t = (j - 80) >> 1;
                if (k < C) v = t == 0 ? (dw_b ? dw_b[k] : 0.f) : t == 1 ? alpha[k] : beta[k];
            }
        } else {  // invert wt16_index for the two halfs of this slot
            _Float16 h[2];
            for (int e2 = 0; e2 < 2; ++e2) {
                const size_t hidx = 2 * (i - d.wt16_off()) + e2;
                const int j = (int)(hidx & 7), m = (int)((hidx >> 3) % d.Mp), g = (int)((hidx >> 3) / d.Mp);
                const int k = (g >> 1) * 16 + (g & 1) * 8 + j;
                h[e2] = (k < C && m < M) ? (_Float16)pw_w[(size_t)m * C + k] : (_Float16)0.f;
            }
            __builtin_memcpy(&v, h, 4);
        }
        packed[i] = v;
    }
}

// ---- two-pass variant -------------------------------------------------------------------------------
// depthwise 5x5 + affine + relu; one workgroup works inside one (b, c) plane so the 28 parameters are
// wave-uniform.  VEC = 4: one thread = 4 cells of a row (G % 4 == 0); VEC = 1: any G.
template <int VEC>
__global__ __launch_bounds__(256) void dw5x5_kernel(const float *__restrict__ x, float *__restrict__ t,
                                                    const float *__restrict__ packed, int C, int G, int bpp) {
    const int pl = blockIdx.x / bpp;
    const int c = pl % C;
    const int local = (blockIdx.x - pl * bpp) * 256 + threadIdx.x;
    const int GV = G / VEC;
    if (local >= G * GV) return;
    const int i = local / GV, jv = local - i * GV;
    const float *xp = x + (size_t)pl * G * G;
    const float *wc = packed + (size_t)(c >> 1) * kCP2 + (c & 1);  // parameter t of this channel at wc[2t]
    float acc[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[q] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 5; ++dy) {
        const int yy = i + dy - 2;
        const bool rok = (unsigned)yy < (unsigned)G;
        const float *row = xp + (size_t)(rok ? yy : 0) * G;
        if constexpr (VEC == 4) {
            float v[12];
            const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 *r4 = reinterpret_cast<const float4 *>(row);
            const bool lok = rok && jv > 0, rok2 = rok && jv + 1 < GV;
            float4 a = r4[lok ? jv - 1 : 0], m = r4[jv], e = r4[rok2 ? jv + 1 : 0];
            a = lok ? a : zero, m = rok ? m : zero, e = rok2 ? e : zero;
            v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = m.x, v[5] = m.y, v[6] = m.z, v[7] = m.w;
            v[8] = e.x, v[9] = e.y, v[10] = e.z, v[11] = e.w;  // cells 4jv-4 .. 4jv+7
#pragma unroll
            for (int dx = 0; dx < 5; ++dx)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = fmaf(wc[2 * (dy * 5 + dx)], v[q + dx + 2], acc[q]);
        } else {
#pragma unroll
            for (int dx = 0; dx < 5; ++dx) {
                const int xx = jv + dx - 2;
                const bool ok = rok && (unsigned)xx < (unsigned)G;
                const float xv = row[ok ? xx : 0];
                acc[0] = fmaf(wc[2 * (dy * 5 + dx)], ok ? xv : 0.f, acc[0]);
            }
        }
    }
    float *dst = t + (size_t)pl * G * G + (size_t)i * G + (size_t)jv * VEC;
#pragma unroll
    for (int q = 0; q < VEC; ++q) dst[q] = dw_finish(acc[q], wc[50], wc[52], wc[54]);
}

// y[b] = W . t[b] + bias on the matrix core; same operands, instruction and k order as the fused kernel.
template <int MT, bool F16>
__global__ __launch_bounds__(256, 2) void pw_gemm_kernel(const float *__restrict__ packed, const float *__restrict__ t,
                                                         float *__restrict__ y, int M, int K, int N) {
    constexpr int BM = 32 * MT;
    __shared__ __attribute__((aligned(16))) float As[kKT][BM];
    __shared__ __attribute__((aligned(16))) float Bs[kKT][kBN];
    const PackDims pd(K, M);
    const float *wt = packed + pd.wt_off();
    const float *bias = packed + pd.bias_off();
    const int Mp = pd.Mp, Kp = pd.Kp;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, m0 = blockIdx.y * BM, n0 = blockIdx.x * kBN;
    const float *tb = t + (size_t)b * K * N;
    const int col = lane & 31, kh = lane >> 5;
    f32x16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int k0 = 0; k0 < Kp; k0 += kKT) {
        for (int e = tid; e < kKT * BM; e += 256) {
            const int k = e / BM, mm = e - k * BM;
            As[k][mm] = m0 + mm < Mp ? wt[pd.wt_index(k0 + k, m0 + mm)] : 0.f;
        }
        for (int e = tid; e < kKT * (kBN / 4); e += 256) {
            const int k = e / (kBN / 4), n4 = e - k * (kBN / 4);
            const int kk = k0 + k, n = n0 + n4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kk < K) {
                const float *src = tb + (size_t)kk * N + n;
                if ((N & 3) == 0) {
                    if (n < N) v = *reinterpret_cast<const float4 *>(src);
                } else {
                    if (n < N) v.x = src[0];
                    if (n + 1 < N) v.y = src[1];
                    if (n + 2 < N) v.z = src[2];
                    if (n + 3 < N) v.w = src[3];
                }
            }
            *reinterpret_cast<float4 *>(&Bs[k][n4 * 4]) = v;
        }
        __syncthreads();
        if constexpr (F16) {
            const _Float16 *w16 = reinterpret_cast<const _Float16 *>(packed + pd.wt16_off());
            f16x8 bv;
#pragma unroll
            for (int j = 0; j < 8; ++j) bv[j] = (_Float16)Bs[8 * kh + j][wave * 32 + col];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int m = m0 + i * 32 + col;
                f16x8 av;
#pragma unroll
                for (int j = 0; j < 8; ++j) av[j] = m < Mp ? w16[pd.wt16_index(k0 + 8 * kh + j, m)] : (_Float16)0.f;
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[i], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < kKT / 2; ++s) {
                const float bv = Bs[2 * s + kh][wave * 32 + col];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const float av = As[2 * s + kh][i * 32 + col];
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    const int n = n0 + wave * 32 + col;
    if (n < N) {
        float *yb = y + (size_t)b * M * N + n;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (m < M) yb[(size_t)m * N] = acc[i][r] + bias[m];
            }
    }
}

// 1x1 conv with a handful of output channels (out_conv: C -> 3): one thread per cell, coalesced along
// the map for every input channel; output channels in groups of 4.
__global__ __launch_bounds__(256) void pw_small_kernel(const float *__restrict__ w, const float *__restrict__ t,
                                                       const float *__restrict__ bias, float *__restrict__ y, int B, int M,
                                                       int K, int N) {
    const long total = (long)B * N;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int n = (int)(idx % N);
        const int b = (int)(idx / N);
        const float *tp = t + (size_t)b * K * N + n;
        for (int mb = 0; mb < M; mb += 4) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < K; ++k) {
                const float v = tp[(size_t)k * N];
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    if (mb + m < M) acc[m] = fmaf(w[(size_t)(mb + m) * K + k], v, acc[m]);
            }
            for (int m = 0; m < 4 && mb + m < M; ++m) y[((size_t)b * M + mb + m) * N + n] = acc[m] + bias[mb + m];
        }
    }
}

inline unsigned grid_for(long total, int cap = 65536) {
    long g = (total + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

// output-channel slabs: as few workgroups along M as possible with <= 7 MFMA row tiles (112 accumulators)
inline void slab_shape(int M, int *nblk, int *mt) {
    const int tiles = (M + 31) / 32;
    *nblk = (tiles + 6) / 7;
    *mt = (tiles + *nblk - 1) / *nblk;
}

}  // namespace

GFN_EXPORT int64_t gfn_conv_block_packed_floats(int C, int M) {
    if (C <= 0 || M <= 0) return 0;
    return (int64_t)PackDims(C, M).total();
}

GFN_EXPORT int gfn_conv_block_pack(const float *dw_w, const float *dw_b, const float *bn_alpha, const float *bn_beta,
                                   const float *pw_w, const float *pw_b, float *packed, int C, int M, gfn_stream_t stream) {
    if (!dw_w || !bn_alpha || !bn_beta || !pw_w || !pw_b || !packed || C <= 0 || M <= 0)
        return gfn::fail(GFN_ERR_INVALID_ARG, "conv_block_pack: bad argument");
    hipLaunchKernelGGL(pack_block_kernel, dim3(grid_for((long)PackDims(C, M).total(), 1024)), dim3(256), 0, (hipStream_t)stream, dw_w,
                       dw_b, bn_alpha, bn_beta, pw_w, pw_b, packed, C, M);
    return gfn::check_launch("pack_block_kernel");
}

GFN_EXPORT int gfn_conv_block_fwd(const float *x, const float *packed, float *y, float *t_scratch, int B, int C, int M, int G,
                                  int variant, gfn_stream_t stream) {
    if (!x || !packed || !y || B < 0 || C <= 0 || M <= 0 || G <= 0) return gfn::fail(GFN_ERR_INVALID_ARG, "conv_block: bad argument");
    if (x == y) return gfn::fail(GFN_ERR_INVALID_ARG, "conv_block: in-place is not supported (cells read their neighbours)");
    if (B == 0) return GFN_OK;
    hipStream_t s = (hipStream_t)stream;
    if ((long)C * G * G > 0x1fffffffL) return gfn::fail(GFN_ERR_INVALID_ARG, "conv_block: a map (C*G*G floats) must stay below 2 GB");
    const int dbg = variant >> 8;  // ablation mask, honoured by -DGFN_ABLATE builds only
#ifdef GFN_ABLATE
    variant &= 0xff;
#endif
    if (variant < 0 || variant > 3) return gfn::fail(GFN_ERR_INVALID_ARG, "conv_block: variant must be 0..3 (got %d)", variant);
    const bool f16 = (variant & 2) != 0;
    const bool fused = !(variant & 1) && (G & 3) == 0;
    if (fused) {
        // tile width: full 128-byte rows where the map allows, narrower tiles for the 5*2^k grids
        if (G % 32 == 0 || G > 160)
            return f16 ? launch_fused<32, true>(x, packed, y, B, M, C, G, dbg, s) : launch_fused<32, false>(x, packed, y, B, M, C, G, dbg, s);
        if (G % 16 == 0 || G > 64)
            return f16 ? launch_fused<16, true>(x, packed, y, B, M, C, G, dbg, s) : launch_fused<16, false>(x, packed, y, B, M, C, G, dbg, s);
        return f16 ? launch_fused<8, true>(x, packed, y, B, M, C, G, dbg, s) : launch_fused<8, false>(x, packed, y, B, M, C, G, dbg, s);
    }
    int nblk, mt;
    slab_shape(M, &nblk, &mt);
    if (!t_scratch) return gfn::fail(GFN_ERR_INVALID_ARG, "conv_block: the two-pass variant needs t_scratch (B*C*G*G floats)");
    if ((long)B * C > 0x7fffff || B > 65535) return gfn::fail(GFN_ERR_INVALID_ARG, "conv_block: batch too large for the two-pass variant");
    const int N = G * G;
    if ((G & 3) == 0) {
        const int bpp = (G * (G / 4) + 255) / 256;
        hipLaunchKernelGGL((dw5x5_kernel<4>), dim3((unsigned)(B * C * bpp)), dim3(256), 0, s, x, t_scratch, packed, C, G, bpp);
    } else {
        const int bpp = (N + 255) / 256;
        hipLaunchKernelGGL((dw5x5_kernel<1>), dim3((unsigned)(B * C * bpp)), dim3(256), 0, s, x, t_scratch, packed, C, G, bpp);
    }
    if (int rc = gfn::check_launch("dw5x5_kernel")) return rc;
    const dim3 grid((N + kBN - 1) / kBN, nblk, B), block(256);
#define GFN_PW(MT)                                                                                                        \
    if (f16)                                                                                                              \
        hipLaunchKernelGGL((pw_gemm_kernel<MT, true>), grid, block, 0, s, packed, (const float *)t_scratch, y, M, C, N);  \
    else                                                                                                                  \
        hipLaunchKernelGGL((pw_gemm_kernel<MT, false>), grid, block, 0, s, packed, (const float *)t_scratch, y, M, C, N)
    switch (mt) {
        case 1: GFN_PW(1); break;
        case 2: GFN_PW(2); break;
        case 3: GFN_PW(3); break;
        case 4: GFN_PW(4); break;
        case 5: GFN_PW(5); break;
        case 6: GFN_PW(6); break;
        default: GFN_PW(7); break;
    }
#undef GFN_PW
    return gfn::check_launch("pw_gemm_kernel");
}

GFN_EXPORT int gfn_pointwise_conv_fwd(const float *w, const float *bias, const float *t, float *y, int B, int M, int K, int N,
                                      gfn_stream_t stream) {
    if (!w || !bias || !t || !y || B < 0 || M <= 0 || K <= 0 || N <= 0) return gfn::fail(GFN_ERR_INVALID_ARG, "pointwise_conv: bad argument");
    if (M > 16) return gfn::fail(GFN_ERR_INVALID_ARG, "pointwise_conv: meant for a few output channels (M <= 16, got %d)", M);
    if (B == 0) return GFN_OK;
    hipLaunchKernelGGL(pw_small_kernel, dim3(grid_for((long)B * N)), dim3(256), 0, (hipStream_t)stream, w, t, bias, y, B, M, K, N);
    return gfn::check_launch("pw_small_kernel");
}
