// corr_softargmax.hip -- global correlation volume and its soft-argmax for gfx950.
//
// Replaces model/network.py:415-428 (GFNet.corr_volume) and :430-440 (GFNet.pos_embed):
//   V[b,j,i]   = sum_c f0[b,c,i] * f1[b,c,j] / sqrt(C)          (stored (B,H1,W1,H0,W0))
//   flow[b,:,i] = sum_j softmax_j(V[b,j,i]) * grid[j],  grid[j] = B-image cell centre (x,y)
// The reference writes the volume (4 MiB per direction at 32^2 x 32^2, 21 MiB at 48^2 x 48^2) and
// reads it back for a softmax over a strided dim.  The fused kernel never writes it: each wave
// owns 32 A-positions (i) and streams all B-positions (j) in tiles of 32 through the exact-fp32
// matrix core (v_mfma_f32_32x32x2_f32: fp32 fmaf chains, two per tile, so no precision is given up),
// with an online softmax (running max / sum / weighted coordinate sums) kept per lane.
//
// MFMA operand mapping (32x32x2 f32, cdna_hip_programming.md section 3): lane l supplies
// A[row=l&31][k=l>>5] and B[k=l>>5][col=l&31]; with rows = j and cols = i both operands are plain
// coalesced loads from the NCHW maps (positions contiguous), no LDS and no transposition.
// Accumulator register r of lane l holds S[j = (r&3)+8(r>>2)+4(l>>5)][i = l&31]: the column
// (i) sits on the lane, so the softmax over j is lane-local apart from one final lane^32 merge.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifndef GFN_CORR_BF16X3
#define GFN_CORR_BF16X3 1   // 0: the row-tile path stays on the exact-fp32 matrix core instruction at every channel count
#endif

// x = h + m + l exactly (three round-to-nearest bf16 pieces of 8 significant bits each cover the 24 of an fp32 value)
__device__ __forceinline__ void split3(float v, __bf16 &h, __bf16 &m, __bf16 &l) {
    h = (__bf16)v;
    const float r1 = v - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

// KS = k-steps of 2 channels held in registers (compile-time so that the operand array stays in
// VGPRs: a runtime-indexed register array would go to scratch).
// FT: storage type of the feature maps (float, or _Float16 for BASELINE config 5: widened on load, products and sums fp32)
template <int KS, bool WRITE_VOL, bool WRITE_FLOW, typename FT>
__global__ __launch_bounds__(256) void corr_softargmax_kernel(const FT *__restrict__ f0, const FT *__restrict__ f1,
                                                              float *__restrict__ vol, float *__restrict__ flow, int B,
                                                              int Bh, int C, int H0, int W0, int H1, int W1, float sqrt_c,
                                                              const bf16x8 *__restrict__ aimg) {
    const int N0 = H0 * W0, N1 = H1 * W1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int itiles = (N0 + 31) >> 5;
    const int wid = blockIdx.x * 4 + wave;
    if (wid >= B * itiles) return;  // no barriers in this kernel
    const int b = wid / itiles, i0 = (wid - b * itiles) << 5;
    const int col = lane & 31, h = lane >> 5;
    const int i = i0 + col;
    const int ic = min(i, N0 - 1);

    // symmetric batches are virtual (Bh = B/2 images per side): direction b >= Bh swaps the roles of
    // the two feature arrays instead of reading a concatenated copy (model/network.py:213-222)
    const FT *f0b = b < Bh ? f0 + (size_t)b * C * N0 : f1 + (size_t)(b - Bh) * C * N0;
    const FT *f1b = b < Bh ? f1 + (size_t)b * C * N1 : f0 + (size_t)(b - Bh) * C * N1;

    // B operand: this wave's 32 columns of f0, all channels, kept in registers
    float bop[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int c = 2 * s + h;
        const float v = (float)f0b[(size_t)min(c, C - 1) * N0 + ic];
        bop[s] = c < C ? v : 0.f;
    }

    const float x_lo = (float)(-1 + 1.0 / W1), x_hi = (float)(1 - 1.0 / W1);
    const float y_lo = (float)(-1 + 1.0 / H1), y_hi = (float)(1 - 1.0 / H1);
    const float inv_w1 = 1.0f / (float)W1;
    float m = -INFINITY, l = 0.f, ax = 0.f, ay = 0.f;
    const float e_scale = 1.4426950408889634f / sqrt_c;    // exp(v / sqrt(C)) = exp2(v * log2(e) / sqrt(C))

    // ---- row tiles (flow only, 32 <= W1 <= 64: the 448 and 672 configurations) ---------------------------------------------
    // A tile of 32 B-positions is (part of) ONE grid row: positions x = 32 p .. 32 p + 31 of row y, those past the row's end
    // masked out (W1 = 48: the second tile of a row is half empty -- a third more matrix work).  Its y coordinate is then
    // wave-uniform, the 16 x coordinates of a lane's accumulator rows are two constant sets, and the 1/sqrt(C) scale folds into
    // the exponent: ~6 VALU instructions per value instead of ~40.  The kernel is VALU-bound (the fp32 MFMA shares the vector
    // ALUs): 48x48 maps, which took the general path below, ran 6.5x longer than 32x32 ones for 2.5x the work.
    if (WRITE_FLOW && !WRITE_VOL && W1 >= 32 && W1 <= 64) {  // wave-uniform
        const int parts = W1 > 32 ? 2 : 1;
        float gxA[16], gxB[16];
        unsigned maskB = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = (r & 3) + 8 * (r >> 2) + 4 * h;
            gxA[r] = gfn::linspace_at(x_lo, x_hi, W1, min(o, W1 - 1));
            gxB[r] = gfn::linspace_at(x_lo, x_hi, W1, min(32 + o, W1 - 1));
            maskB |= (32 + o < W1 ? 1u : 0u) << r;
        }
        auto load_row_tile = [&](float (&a)[KS], int t) {
            const int y = parts == 1 ? t : t >> 1, pp = parts == 1 ? 0 : t & 1;
            const int jl = y * W1 + min(32 * pp + col, W1 - 1);
#pragma unroll
            for (int s = 0; s < KS; ++s) a[s] = (float)f1b[(size_t)min(2 * s + h, C - 1) * N1 + jl];  // channels >= C meet a zero in bop
        };
        auto softmax_tile = [&](f32x16 acc, const float (&gx)[16], int y) {
            float mt = acc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mt = fmaxf(mt, acc[r]);
            const float mn = fmaxf(m, mt);
            const float sc = __builtin_amdgcn_exp2f((m - mn) * e_scale);  // m = -inf on the first tile -> 0
            float lt = 0.f, axt = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f((acc[r] - mn) * e_scale);  // masked positions: exp2(-inf) = 0
                lt += e;
                axt = fmaf(e, gx[r], axt);
            }
            const float gy = gfn::linspace_at(y_lo, y_hi, H1, y);
            l = fmaf(l, sc, lt);
            ax = fmaf(ax, sc, axt);
            ay = fmaf(ay, sc, lt * gy);
            m = mn;
        };
        const int ntiles = H1 * parts;
        if constexpr (KS == 32 && GFN_CORR_BF16X3 != 0) {
            // Round 6, 64-channel maps (GFNet's stride-16 features): the products on the bf16 matrix core instruction with both operands
            // split three ways (x = h + m + l, exact).  Six of the nine piece products are kept -- h.h, h.m, m.h, m.m, h.l, l.h; the dropped
            // m.l, l.m, l.l are <= 2^-23 of |a||b|, the size of an fp32 product's own rounding -- in 24 v_mfma_f32_32x32x16_bf16 per tile
            // (768 matrix cycles) instead of 32 v_mfma_f32_32x32x2_f32 (2 048), and unlike the fp32 instruction they leave the vector ALUs
            // to the softmax and to the next tile's splitting.  Same accumulator layout, same softmax.  Lane l supplies row / column
            // l & 31 and channels 16 c + 8 (l >> 5) + e, e = 0..7, of chunk c: plain coalesced NCHW loads as before.
            bf16x8 bh[4], bm[4], bl[4];
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = 16 * c4 + 8 * h + e;
                    const float v = c < C ? (float)f0b[(size_t)min(c, C - 1) * N0 + ic] : 0.f;
                    __bf16 ph, pm, pl;
                    split3(v, ph, pm, pl);
                    bh[c4][e] = ph; bm[c4][e] = pm; bl[c4][e] = pl;
                }
            auto load_raw = [&](float (&a)[32], int t) {
                const int y = parts == 1 ? t : t >> 1, pp = parts == 1 ? 0 : t & 1;
                const int jl = y * W1 + min(32 * pp + col, W1 - 1);
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
                    for (int e = 0; e < 8; ++e) a[8 * c4 + e] = (float)f1b[(size_t)min(16 * c4 + 8 * h + e, C - 1) * N1 + jl];  // channels >= C meet zeros in b*
            };
            if (aimg) {  // (kernel argument: uniform)
                // the B-positions' operand comes pre-split from the workspace (split_rows_kernel below): every one of the 32 waves that
                // walk a direction's tiles used to split the same values again -- ~300 of a tile's ~420 vector instructions, on one
                // wave per SIMD.  Twelve 16-byte loads per tile and lane instead of 32 4-byte ones, nothing but the softmax on the VALU.
                const bf16x8 *img = aimg + (size_t)b * ntiles * 12 * 64 + lane;
                bf16x8 a_cur[12], a_nxt[12];
#pragma unroll
                for (int k = 0; k < 12; ++k) a_cur[k] = img[k * 64];
#pragma unroll
                for (int k = 0; k < 12; ++k) asm volatile("" : "+v"(a_cur[k]));
                for (int t = 0; t < ntiles; ++t) {
                    if (t + 1 < ntiles) {
#pragma unroll
                        for (int k = 0; k < 12; ++k) a_nxt[k] = img[((size_t)(t + 1) * 12 + k) * 64];
                    }
                    f32x16 accs = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc = accs;
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) {
                        accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[4 + c4], bm[c4], accs, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[c4], bm[c4], acc, 0, 0, 0);
                    }
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) {
                        accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[c4], bl[c4], accs, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[4 + c4], bh[c4], acc, 0, 0, 0);
                    }
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) {
                        accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[8 + c4], bh[c4], accs, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[c4], bh[c4], acc, 0, 0, 0);
                    }
                    acc += accs;
#pragma unroll
                    for (int k = 0; k < 12; ++k) a_cur[k] = a_nxt[k];
                    if (parts == 1 || !(t & 1)) {
                        softmax_tile(acc, gxA, parts == 1 ? t : t >> 1);
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[r] = ((maskB >> r) & 1u) ? acc[r] : -INFINITY;
                        softmax_tile(acc, gxB, t >> 1);
                    }
                }
            } else {
            float r_cur[32], r_nxt[32];
            load_raw(r_cur, 0);
#pragma unroll
            for (int k = 0; k < 32; ++k) asm volatile("" : "+v"(r_cur[k]));   // land the first tile before the loop (see below)
            for (int t = 0; t < ntiles; ++t) {
                if (t + 1 < ntiles) load_raw(r_nxt, t + 1);
                bf16x8 ah[4], am[4], al[4];
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        __bf16 ph, pm, pl;
                        split3(r_cur[8 * c4 + e], ph, pm, pl);
                        ah[c4][e] = ph; am[c4][e] = pm; al[c4][e] = pl;
                    }
                // two chains: the small classes and the large ones, smallest terms first inside each; summed at the end
                f32x16 accs = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc = accs;
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[c4], bm[c4], accs, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[c4], bm[c4], acc, 0, 0, 0);
                }
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[c4], bl[c4], accs, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[c4], bh[c4], acc, 0, 0, 0);
                }
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[c4], bh[c4], accs, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[c4], bh[c4], acc, 0, 0, 0);
                }
                acc += accs;
#pragma unroll
                for (int k = 0; k < 32; ++k) r_cur[k] = r_nxt[k];
                if (parts == 1 || !(t & 1)) {
                    softmax_tile(acc, gxA, parts == 1 ? t : t >> 1);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = ((maskB >> r) & 1u) ? acc[r] : -INFINITY;
                    softmax_tile(acc, gxB, t >> 1);
                }
            }
            }
        } else {
        float a_cur[KS], a_nxt[KS];
        load_row_tile(a_cur, 0);
        // land the first tile before the loop: otherwise the wait-count pass assumes 64 loads in flight at the loop head and
        // makes every MFMA of every tile wait for the *prefetch* it was meant to overlap with
#pragma unroll
        for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(a_cur[s]));
        for (int t = 0; t < ntiles; ++t) {
            if (t + 1 < ntiles) load_row_tile(a_nxt, t + 1);
            // two independent accumulation chains (even / odd k-steps): a single chain left the matrix pipe waiting on its own
            // result between issues (0.152 -> 0.122 ms for 64 directions; four chains: 0.131); summed at the end
            f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc2 = acc;
#pragma unroll
            for (int s = 0; s < KS; s += 2) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[s], bop[s], acc, 0, 0, 0);
                if (s + 1 < KS) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[s + 1], bop[s + 1], acc2, 0, 0, 0);
            }
            acc += acc2;
#pragma unroll
            for (int s = 0; s < KS; ++s) a_cur[s] = a_nxt[s];
            if (parts == 1 || !(t & 1)) {
                softmax_tile(acc, gxA, parts == 1 ? t : t >> 1);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = ((maskB >> r) & 1u) ? acc[r] : -INFINITY;
                softmax_tile(acc, gxB, t >> 1);
            }
        }
        }
        m = m / sqrt_c;  // the running maximum was kept in unscaled units
        const float m2 = __shfl_xor(m, 32), l2 = __shfl_xor(l, 32), ax2 = __shfl_xor(ax, 32), ay2 = __shfl_xor(ay, 32);
        const float mn = fmaxf(m, m2);
        const float s1 = __expf(m - mn), s2 = __expf(m2 - mn);
        const float lt = l * s1 + l2 * s2;
        const float fx = (ax * s1 + ax2 * s2) / lt, fy = (ay * s1 + ay2 * s2) / lt;
        if (h == 0 && i < N0) {
            flow[((size_t)b * 2 + 0) * N0 + i] = fx;
            flow[((size_t)b * 2 + 1) * N0 + i] = fy;
        }
        return;
    }

    // A operand of one tile (32 B-positions x all channels): loads are branch-free (clamped address) so the
    // compiler batches them, and the next tile's operand is requested before this tile's MFMA chain starts -- with two
    // waves per SIMD nothing else hides the L2 latency.
    auto load_tile = [&](float (&a)[KS], int j0) {
        const int jl = min(j0 + col, N1 - 1);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            a[s] = (float)f1b[(size_t)min(2 * s + h, C - 1) * N1 + jl];  // channels >= C meet a zero in bop
        }
    };
    float a_cur[KS], a_nxt[KS];
    load_tile(a_cur, 0);
    // land the first tile before the loop: otherwise the wait-count pass assumes 64 loads in flight at the loop head and
    // makes every MFMA of every tile wait for the *prefetch* it was meant to overlap with
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(a_cur[s]));
    for (int j0 = 0; j0 < N1; j0 += 32) {
        if (j0 + 32 < N1) load_tile(a_nxt, j0 + 32);
        // two independent accumulation chains (even / odd k-steps): a single chain left the matrix pipe waiting on its own
        // result between issues (0.152 -> 0.122 ms for 64 directions; four chains: 0.131); summed at the end
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc2 = acc;
#pragma unroll
        for (int s = 0; s < KS; s += 2) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[s], bop[s], acc, 0, 0, 0);
            if (s + 1 < KS) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[s + 1], bop[s + 1], acc2, 0, 0, 0);
        }
        acc += acc2;
#pragma unroll
        for (int s = 0; s < KS; ++s) a_cur[s] = a_nxt[s];
        float sv[16];
        float mt = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const float v = acc[r] / sqrt_c;
            if (WRITE_VOL) {
                if (j < N1 && i < N0) vol[((size_t)b * N1 + j) * N0 + i] = v;
            }
            sv[r] = (j < N1) ? v : -INFINITY;
            mt = fmaxf(mt, sv[r]);
        }
        if (WRITE_FLOW) {
            const float mn = fmaxf(m, mt);
            const float sc = __expf(m - mn);  // m = -inf on the first tile -> 0
            l *= sc; ax *= sc; ay *= sc;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int jy = (int)(((float)j + 0.5f) * inv_w1);
                const int jx = j - jy * W1;
                const float e = __expf(sv[r] - mn);  // exp(-inf) = 0 for the padded rows
                l += e;
                ax = fmaf(e, gfn::linspace_at(x_lo, x_hi, W1, jx), ax);
                ay = fmaf(e, gfn::linspace_at(y_lo, y_hi, H1, min(jy, H1 - 1)), ay);
            }
            m = mn;
        }
    }
    if (WRITE_FLOW) {
        // merge the two half-waves (same column i, disjoint rows j)
        const float m2 = __shfl_xor(m, 32), l2 = __shfl_xor(l, 32), ax2 = __shfl_xor(ax, 32), ay2 = __shfl_xor(ay, 32);
        const float mn = fmaxf(m, m2);
        const float s1 = __expf(m - mn), s2 = __expf(m2 - mn);
        const float lt = l * s1 + l2 * s2;
        const float fx = (ax * s1 + ax2 * s2) / lt, fy = (ay * s1 + ay2 * s2) / lt;
        if (h == 0 && i < N0) {
            flow[((size_t)b * 2 + 0) * N0 + i] = fx;
            flow[((size_t)b * 2 + 1) * N0 + i] = fy;
        }
    }
}

// The row-tile path on pre-split operand images as a kernel of its own (round 6): nothing of the general kernel's other paths lives in
// its register allocation, so two waves fit a SIMD (<= 256 registers; the general kernel sits at 292 = one wave) and the image loads of
// one wave hide behind the other's matrix work.  PARTS = row tiles per grid row (1: W1 == 32, 2: 33..64).
template <int PARTS, typename FT>
__global__ __launch_bounds__(256, 2) void corr_softargmax_img_kernel(const FT *__restrict__ f0, const FT *__restrict__ f1, float *__restrict__ flow,
                                                                     int B, int Bh, int C, int H0, int W0, int H1, int W1, float sqrt_c,
                                                                     const bf16x8 *__restrict__ aimg) {
    const int N0 = H0 * W0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int itiles = (N0 + 31) >> 5;
    const int wid = blockIdx.x * 4 + wave;
    if (wid >= B * itiles) return;
    const int b = wid / itiles, i0 = (wid - b * itiles) << 5;
    const int col = lane & 31, h = lane >> 5;
    const int i = i0 + col;
    const int ic = min(i, N0 - 1);
    const FT *f0b = b < Bh ? f0 + (size_t)b * C * N0 : f1 + (size_t)(b - Bh) * C * N0;
    bf16x8 bh[4], bm[4], bl[4];
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = 16 * c4 + 8 * h + e;
            const float v = c < C ? (float)f0b[(size_t)min(c, C - 1) * N0 + ic] : 0.f;
            __bf16 ph, pm, pl;
            split3(v, ph, pm, pl);
            bh[c4][e] = ph; bm[c4][e] = pm; bl[c4][e] = pl;
        }
    const float x_lo = (float)(-1 + 1.0 / W1), x_hi = (float)(1 - 1.0 / W1);
    const float y_lo = (float)(-1 + 1.0 / H1), y_hi = (float)(1 - 1.0 / H1);
    float m = -INFINITY, l = 0.f, ax = 0.f, ay = 0.f;
    const float e_scale = 1.4426950408889634f / sqrt_c;
    float gxA[16], gxB[PARTS == 2 ? 16 : 1];
    unsigned maskB = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = (r & 3) + 8 * (r >> 2) + 4 * h;
        gxA[r] = gfn::linspace_at(x_lo, x_hi, W1, min(o, W1 - 1));
        if (PARTS == 2) {
            gxB[r] = gfn::linspace_at(x_lo, x_hi, W1, min(32 + o, W1 - 1));
            maskB |= (32 + o < W1 ? 1u : 0u) << r;
        }
    }
    auto softmax_tile = [&](f32x16 acc, const float *gx, int y) {   // as in corr_softargmax_kernel
        float mt = acc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, acc[r]);
        const float mn = fmaxf(m, mt);
        const float sc = __builtin_amdgcn_exp2f((m - mn) * e_scale);
        float lt = 0.f, axt = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f((acc[r] - mn) * e_scale);
            lt += e;
            axt = fmaf(e, gx[r], axt);
        }
        const float gy = gfn::linspace_at(y_lo, y_hi, H1, y);
        l = fmaf(l, sc, lt);
        ax = fmaf(ax, sc, axt);
        ay = fmaf(ay, sc, lt * gy);
        m = mn;
    };
    const int ntiles = H1 * PARTS;
    const bf16x8 *img = aimg + (size_t)b * ntiles * 12 * 64 + lane;
    // (issuing the products of tile t + 1 in front of the softmax of tile t -- a software pipeline over the tiles, 238 registers -- measured
    // SLOWER: 72 against 64 us; the second wave of the SIMD already fills the matrix pipe under this wave's exponentials)
    bf16x8 a_cur[12], a_nxt[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) a_cur[k] = img[k * 64];
#pragma unroll
    for (int k = 0; k < 12; ++k) asm volatile("" : "+v"(a_cur[k]));   // land the first tile before the loop (see corr_softargmax_kernel)
    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) {
#pragma unroll
            for (int k = 0; k < 12; ++k) a_nxt[k] = img[((size_t)(t + 1) * 12 + k) * 64];
        }
        // two chains: the small classes and the large ones, smallest terms first inside each; summed at the end
        f32x16 accs = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc = accs;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[4 + c4], bm[c4], accs, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[c4], bm[c4], acc, 0, 0, 0);
        }
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[c4], bl[c4], accs, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[4 + c4], bh[c4], acc, 0, 0, 0);
        }
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[8 + c4], bh[c4], accs, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[c4], bh[c4], acc, 0, 0, 0);
        }
        acc += accs;
#pragma unroll
        for (int k = 0; k < 12; ++k) a_cur[k] = a_nxt[k];
        if (PARTS == 1 || !(t & 1)) {
            softmax_tile(acc, gxA, PARTS == 1 ? t : t >> 1);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = ((maskB >> r) & 1u) ? acc[r] : -INFINITY;
            softmax_tile(acc, gxB, t >> 1);
        }
    }
    m = m / sqrt_c;  // the running maximum was kept in unscaled units
    const float m2 = __shfl_xor(m, 32), l2 = __shfl_xor(l, 32), ax2 = __shfl_xor(ax, 32), ay2 = __shfl_xor(ay, 32);
    const float mn = fmaxf(m, m2);
    const float s1 = __expf(m - mn), s2 = __expf(m2 - mn);
    const float lt = l * s1 + l2 * s2;
    const float fx = (ax * s1 + ax2 * s2) / lt, fy = (ay * s1 + ay2 * s2) / lt;
    if (h == 0 && i < N0) {
        flow[((size_t)b * 2 + 0) * N0 + i] = fx;
        flow[((size_t)b * 2 + 1) * N0 + i] = fy;
    }
}

// Workspace of the split-bf16 row-tile path: the B-positions' operand of every direction, split once.  Image of direction b, row tile t
// (32 positions of one grid row, as corr_softargmax_kernel walks them), piece p (h, m, l), 16-channel chunk c4: 64 lanes x 16 bytes, lane
// (h = lane >> 5, col = lane & 31) holding channels 16 c4 + 8 h + e, e = 0..7, of position y W1 + min(32 part + col, W1 - 1).
template <typename FT>
__global__ __launch_bounds__(256) void split_rows_kernel(const FT *__restrict__ f0, const FT *__restrict__ f1, bf16x8 *__restrict__ img, int B,
                                                         int Bh, int C, int H1, int W1) {
    const int parts = W1 > 32 ? 2 : 1, ntiles = H1 * parts, N1 = H1 * W1;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;   // (b, t, c4, lane)
    if (idx >= (long)B * ntiles * 4 * 64) return;
    const int lane = (int)(idx & 63), c4 = (int)((idx >> 6) & 3);
    const long bt = idx >> 8;
    const int t = (int)(bt % ntiles), b = (int)(bt / ntiles);
    const int col = lane & 31, h = lane >> 5;
    const FT *f1b = b < Bh ? f1 + (size_t)b * C * N1 : f0 + (size_t)(b - Bh) * C * N1;  // (symmetric: equal map sizes, checked by the caller)
    const int y = parts == 1 ? t : t >> 1, pp = parts == 1 ? 0 : t & 1;
    const int jl = y * W1 + min(32 * pp + col, W1 - 1);
    bf16x8 ph, pm, pl;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = 16 * c4 + 8 * h + e;
        const float v = c < C ? (float)f1b[(size_t)c * N1 + jl] : 0.f;
        __bf16 a, m_, l_;
        split3(v, a, m_, l_);
        ph[e] = a; pm[e] = m_; pl[e] = l_;
    }
    bf16x8 *dst = img + ((size_t)b * ntiles + t) * 12 * 64 + lane;
    dst[(0 + c4) * 64] = ph;
    dst[(4 + c4) * 64] = pm;
    dst[(8 + c4) * 64] = pl;
}

// bytes of workspace the split-bf16 path wants for this shape (0: the shape does not take it)
int64_t split_ws_bytes(int B, int C, int H1, int W1) {
#ifndef GFN_CORR_PRESPLIT
#define GFN_CORR_PRESPLIT 1   // 0: A/B builds in which every wave splits the operand itself
#endif
    if (GFN_CORR_BF16X3 == 0 || GFN_CORR_PRESPLIT == 0 || C <= 32 || C > 64 || W1 < 32 || W1 > 64) return 0;
    const int64_t ntiles = (int64_t)H1 * (W1 > 32 ? 2 : 1);
    return (int64_t)B * ntiles * 12 * 64 * 16;
}

// pos_embed on an explicit volume (model/network.py:430-440): one thread per (b, i), coalesced
// over i, online softmax over j.  Only used when a caller hands in a volume of its own.
__global__ __launch_bounds__(256) void pos_embed_kernel(const float *__restrict__ vol, float *__restrict__ flow, int B,
                                                        int H0, int W0, int H1, int W1) {
    const int N0 = H0 * W0, N1 = H1 * W1;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * N0) return;
    const int b = (int)(idx / N0), i = (int)(idx - (long)b * N0);
    const float x_lo = (float)(-1 + 1.0 / W1), x_hi = (float)(1 - 1.0 / W1);
    const float y_lo = (float)(-1 + 1.0 / H1), y_hi = (float)(1 - 1.0 / H1);
    const float *v = vol + (size_t)b * N1 * N0 + i;
    float m = -INFINITY, l = 0.f, ax = 0.f, ay = 0.f;
    int jx = 0, jy = 0;
    for (int j = 0; j < N1; ++j) {
        const float s = v[(size_t)j * N0];
        if (s > m) {
            const float sc = __expf(m - s);
            l *= sc; ax *= sc; ay *= sc;
            m = s;
        }
        const float e = __expf(s - m);
        l += e;
        ax = fmaf(e, gfn::linspace_at(x_lo, x_hi, W1, jx), ax);
        ay = fmaf(e, gfn::linspace_at(y_lo, y_hi, H1, jy), ay);
        if (++jx == W1) { jx = 0; ++jy; }
    }
    flow[((size_t)b * 2 + 0) * N0 + i] = ax / l;
    flow[((size_t)b * 2 + 1) * N0 + i] = ay / l;
}

int check_args(const void *f0, const void *f1, int B, int C, int H0, int W0, int H1, int W1) {
    if (!f0 || !f1) return gfn::fail(GFN_ERR_INVALID_ARG, "corr: null feature pointer");
    if (B < 0 || C <= 0 || H0 <= 0 || W0 <= 0 || H1 <= 0 || W1 <= 0)
        return gfn::fail(GFN_ERR_INVALID_ARG, "corr: bad size B=%d C=%d %dx%d vs %dx%d", B, C, H0, W0, H1, W1);
    if (C > 128) return gfn::fail(GFN_ERR_INVALID_ARG, "corr: C=%d > 128 channels not supported", C);
    if ((long)H0 * W0 >= (1L << 24) || (long)H1 * W1 >= (1L << 24))
        return gfn::fail(GFN_ERR_INVALID_ARG, "corr: map too large");
    return GFN_OK;
}

template <bool WV, bool WF, typename FT>
int launch_corr(const FT *f0, const FT *f1, float *vol, float *flow, int B, int Bh, int C, int H0, int W0, int H1,
                int W1, hipStream_t stream, void *ws = nullptr, int64_t ws_bytes = 0) {
    const int waves = B * ((H0 * W0 + 31) / 32);
    const dim3 grid((waves + 3) / 4), block(256);
    const float sc = (float)sqrt((double)C);
    const bf16x8 *aimg = nullptr;
    if (WF && !WV) {
        const int64_t need = split_ws_bytes(B, C, H1, W1);
        if (need > 0 && ws && ws_bytes >= need && ((uintptr_t)ws & 15) == 0) {   // without a workspace the kernel splits the operand itself
            const long items = (long)B * H1 * (W1 > 32 ? 2 : 1) * 4 * 64;
            hipLaunchKernelGGL((split_rows_kernel<FT>), dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream, f0, f1,
                               static_cast<bf16x8 *>(ws), B, Bh, C, H1, W1);
            if (int e = gfn::check_launch("split_rows_kernel")) return e;
            aimg = static_cast<const bf16x8 *>(ws);
        }
    }
    if constexpr (WF && !WV) {
        if (aimg) {
            if (W1 > 32)
                hipLaunchKernelGGL((corr_softargmax_img_kernel<2, FT>), grid, block, 0, stream, f0, f1, flow, B, Bh, C, H0, W0, H1, W1, sc, aimg);
            else
                hipLaunchKernelGGL((corr_softargmax_img_kernel<1, FT>), grid, block, 0, stream, f0, f1, flow, B, Bh, C, H0, W0, H1, W1, sc, aimg);
            return gfn::check_launch("corr_softargmax_img_kernel");
        }
    }
    if (C <= 16)
        hipLaunchKernelGGL((corr_softargmax_kernel<8, WV, WF, FT>), grid, block, 0, stream, f0, f1, vol, flow, B, Bh, C, H0, W0, H1, W1, sc, aimg);
    else if (C <= 32)
        hipLaunchKernelGGL((corr_softargmax_kernel<16, WV, WF, FT>), grid, block, 0, stream, f0, f1, vol, flow, B, Bh, C, H0, W0, H1, W1, sc, aimg);
    else if (C <= 64)
        hipLaunchKernelGGL((corr_softargmax_kernel<32, WV, WF, FT>), grid, block, 0, stream, f0, f1, vol, flow, B, Bh, C, H0, W0, H1, W1, sc, aimg);
    else
        hipLaunchKernelGGL((corr_softargmax_kernel<64, WV, WF, FT>), grid, block, 0, stream, f0, f1, vol, flow, B, Bh, C, H0, W0, H1, W1, sc, aimg);
    return gfn::check_launch("corr_softargmax_kernel");
}

}  // namespace

GFN_EXPORT int gfn_corr_softargmax_fwd(const float *f0, const float *f1, float *flow, int B, int C, int H0, int W0,
                                       int H1, int W1, int symmetric, gfn_stream_t stream) {
    if (int e = check_args(f0, f1, B, C, H0, W0, H1, W1)) return e;
    if (!flow) return gfn::fail(GFN_ERR_INVALID_ARG, "corr_softargmax: null flow");
    if (symmetric && ((B & 1) || H0 != H1 || W0 != W1))
        return gfn::fail(GFN_ERR_INVALID_ARG, "corr_softargmax: symmetric needs an even batch and equal map sizes");
    if (B == 0) return GFN_OK;
    return launch_corr<false, true>(f0, f1, nullptr, flow, B, symmetric ? B / 2 : B, C, H0, W0, H1, W1, (hipStream_t)stream);
}

GFN_EXPORT int gfn_corr_softargmax_fwd_dt(const void *f0, const void *f1, int dtype, float *flow, int B, int C, int H0, int W0,
                                          int H1, int W1, int symmetric, gfn_stream_t stream) {
    if (dtype == GFN_F32)
        return gfn_corr_softargmax_fwd(static_cast<const float *>(f0), static_cast<const float *>(f1), flow, B, C, H0, W0, H1, W1, symmetric, stream);
    if (dtype != GFN_F16) return gfn::fail(GFN_ERR_INVALID_ARG, "corr_softargmax: feature dtype must be GFN_F32 or GFN_F16");
    if (int e = check_args(static_cast<const float *>(f0), static_cast<const float *>(f1), B, C, H0, W0, H1, W1)) return e;
    if (!flow) return gfn::fail(GFN_ERR_INVALID_ARG, "corr_softargmax: null flow");
    if (symmetric && ((B & 1) || H0 != H1 || W0 != W1))
        return gfn::fail(GFN_ERR_INVALID_ARG, "corr_softargmax: symmetric needs an even batch and equal map sizes");
    if (B == 0) return GFN_OK;
    return launch_corr<false, true>(static_cast<const _Float16 *>(f0), static_cast<const _Float16 *>(f1), nullptr, flow, B, symmetric ? B / 2 : B, C,
                                    H0, W0, H1, W1, (hipStream_t)stream);
}

GFN_EXPORT int64_t gfn_corr_softargmax_ws_bytes(int B, int C, int H1, int W1) { return split_ws_bytes(B, C, H1, W1); }

GFN_EXPORT int gfn_corr_softargmax_fwd_ws(const void *f0, const void *f1, int dtype, float *flow, int B, int C, int H0, int W0, int H1,
                                          int W1, int symmetric, void *ws, int64_t ws_bytes, gfn_stream_t stream) {
    if (dtype != GFN_F32 && dtype != GFN_F16) return gfn::fail(GFN_ERR_INVALID_ARG, "corr_softargmax: feature dtype must be GFN_F32 or GFN_F16");
    if (int e = check_args(f0, f1, B, C, H0, W0, H1, W1)) return e;
    if (!flow) return gfn::fail(GFN_ERR_INVALID_ARG, "corr_softargmax: null flow");
    if (symmetric && ((B & 1) || H0 != H1 || W0 != W1))
        return gfn::fail(GFN_ERR_INVALID_ARG, "corr_softargmax: symmetric needs an even batch and equal map sizes");
    if (B == 0) return GFN_OK;
    const int Bh = symmetric ? B / 2 : B;
    if (dtype == GFN_F16)
        return launch_corr<false, true>(static_cast<const _Float16 *>(f0), static_cast<const _Float16 *>(f1), nullptr, flow, B, Bh, C, H0, W0, H1, W1,
                                        (hipStream_t)stream, ws, ws_bytes);
    return launch_corr<false, true>(static_cast<const float *>(f0), static_cast<const float *>(f1), nullptr, flow, B, Bh, C, H0, W0, H1, W1,
                                    (hipStream_t)stream, ws, ws_bytes);
}

GFN_EXPORT int gfn_corr_volume_fwd(const float *f0, const float *f1, float *vol, float *flow_or_null, int B, int C,
                                   int H0, int W0, int H1, int W1, gfn_stream_t stream) {
    if (int e = check_args(f0, f1, B, C, H0, W0, H1, W1)) return e;
    if (!vol) return gfn::fail(GFN_ERR_INVALID_ARG, "corr_volume: null volume");
    if (B == 0) return GFN_OK;
    if (flow_or_null) return launch_corr<true, true>(f0, f1, vol, flow_or_null, B, B, C, H0, W0, H1, W1, (hipStream_t)stream);
    return launch_corr<true, false>(f0, f1, vol, nullptr, B, B, C, H0, W0, H1, W1, (hipStream_t)stream);
}

GFN_EXPORT int gfn_pos_embed_fwd(const float *vol, float *flow, int B, int H0, int W0, int H1, int W1,
                                 gfn_stream_t stream) {
    if (!vol || !flow || B < 0 || H0 <= 0 || W0 <= 0 || H1 <= 0 || W1 <= 0)
        return gfn::fail(GFN_ERR_INVALID_ARG, "pos_embed: bad argument");
    if (B == 0) return GFN_OK;
    const long total = (long)B * H0 * W0;
    hipLaunchKernelGGL(pos_embed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, vol,
                       flow, B, H0, W0, H1, W1);
    return gfn::check_launch("pos_embed_kernel");
}
