// refiner_input.h -- the body of the refiner-input kernel (ConvRefiner.forward prefix, model/network.py:533-555), shared by
// grid_ops.hip (gfn_refiner_input_fwd: the kernel on its own) and local_corr.hip (gfn_refiner_input_plan_fwd: the same launch
// also plans the tiles of the local correlation that follows, so that call needs no plan launch of its own).
#pragma once
#include "common.h"

namespace gfn_ri {

__device__ __forceinline__ float unnorm(float g, int size) { return ((g + 1.f) * (float)size - 1.f) / 2.f; }

struct Bilin {
    int x0, y0;
    float w00, w01, w10, w11;
    bool xa, xb, ya, yb;
};

// grid_sample's bilinear set-up (ATen grid_sampler_2d): corners nw,ne,sw,se; zeros padding.
__device__ __forceinline__ Bilin bilin_setup(float gx, float gy, int W, int H) {
    Bilin s;
    const float ix = unnorm(gx, W), iy = unnorm(gy, H);
    const float fx = floorf(ix), fy = floorf(iy);
    const bool sane = (fx > -1e6f) & (fx < 1e6f) & (fy > -1e6f) & (fy < 1e6f);
    s.x0 = sane ? (int)fx : -4;
    s.y0 = sane ? (int)fy : -4;
    s.w00 = (fx + 1.f - ix) * (fy + 1.f - iy);
    s.w01 = (ix - fx) * (fy + 1.f - iy);
    s.w10 = (fx + 1.f - ix) * (iy - fy);
    s.w11 = (ix - fx) * (iy - fy);
    s.xa = (unsigned)s.x0 < (unsigned)W;
    s.xb = (unsigned)(s.x0 + 1) < (unsigned)W;
    s.ya = (unsigned)s.y0 < (unsigned)H;
    s.yb = (unsigned)(s.y0 + 1) < (unsigned)H;
    return s;
}

__device__ __forceinline__ float bilin_fetch(const float *pl, int W, const Bilin &s) {
    const long o = (long)s.y0 * W + s.x0;
    float v = 0.f;
    if (s.ya & s.xa) v += pl[o] * s.w00;
    if (s.ya & s.xb) v += pl[o + 1] * s.w01;
    if (s.yb & s.xa) v += pl[o + W] * s.w10;
    if (s.yb & s.xb) v += pl[o + W + 1] * s.w11;
    return v;
}

// Clamped form of a bilinear set-up: four always-valid offsets and four weights that are zero for
// out-of-image corners, so the gathers need no branches (a zero weight times any finite value adds
// an exact 0; corner order nw, ne, sw, se is kept).
struct BilinC {
    int o[4];
    float w[4];
};

__device__ __forceinline__ BilinC bilin_clamped(float gx, float gy, int W, int H) {
    const Bilin s = bilin_setup(gx, gy, W, H);
    BilinC c;
    const int o00 = s.y0 * W + s.x0;
    c.o[0] = (s.ya & s.xa) ? o00 : 0;
    c.o[1] = (s.ya & s.xb) ? o00 + 1 : 0;
    c.o[2] = (s.yb & s.xa) ? o00 + W : 0;
    c.o[3] = (s.yb & s.xb) ? o00 + W + 1 : 0;
    c.w[0] = (s.ya & s.xa) ? s.w00 : 0.f;
    c.w[1] = (s.ya & s.xb) ? s.w01 : 0.f;
    c.w[2] = (s.yb & s.xa) ? s.w10 : 0.f;
    c.w[3] = (s.yb & s.xb) ? s.w11 : 0.f;
    return c;
}

// Pair form of a bilinear set-up: the two corners of an image row are adjacent, so one 8-byte gather (4-byte aligned)
// fetches both.  o[0]/o[1] = offsets of the pair in rows y0 / y0+1, always inside the map (column clamped to 0..W-2, an
// out-of-image row reads row 0); w = weights of (row y0: left, right; row y0+1: left, right) *of the fetched pixels*:
// zero for a fetched pixel that is not the corner it stands in for, so out-of-image corners add an exact 0 and the
// corner order nw, ne, sw, se of grid_sample is kept.
struct BilinP {
    unsigned o[2];
    float w[4];
};
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));

__device__ __forceinline__ BilinP bilin_pairs(float gx, float gy, int W, int H) {
    const Bilin s = bilin_setup(gx, gy, W, H);
    BilinP c;
    const int ox = min(max(s.x0, 0), W - 2);
    // fetched left pixel = column ox, right = ox + 1; corner columns are x0 (weights w*0) and x0+1 (weights w*1)
    const float l0 = (s.xa & (s.x0 == ox)) ? 1.f : 0.f, l1 = (s.xb & (s.x0 + 1 == ox)) ? 1.f : 0.f;      // left stands for x0 / x0+1
    const float r0 = (s.xa & (s.x0 == ox + 1)) ? 1.f : 0.f, r1 = (s.xb & (s.x0 + 1 == ox + 1)) ? 1.f : 0.f;  // right stands for x0 / x0+1
    const float ta = s.ya ? 1.f : 0.f, tb = s.yb ? 1.f : 0.f;
    c.o[0] = (unsigned)((s.ya ? s.y0 : 0) * W + ox);
    c.o[1] = (unsigned)((s.yb ? s.y0 + 1 : 0) * W + ox);
    // exactly one of (l0, l1) and one of (r0, r1) can be 1, so each product below is w or 0 -- no rounding added
    c.w[0] = ta * (l0 * s.w00 + l1 * s.w01);
    c.w[1] = ta * (r0 * s.w00 + r1 * s.w01);
    c.w[2] = tb * (l0 * s.w10 + l1 * s.w11);
    c.w[3] = tb * (r0 * s.w10 + r1 * s.w11);
    return c;
}

// One thread per (direction, grid cell): both bilinear set-ups once, then the channels in groups of
// 8 with all 64 gathers of a group in flight; stores run along the grid row for every channel.
// A workgroup never straddles two directions (blockIdx.y = direction), so every plane base is a scalar and a load is
// "scalar base + 32-bit lane offset": no 64-bit address arithmetic per gather (it was 2 of every 5 vector instructions
// and pushed the kernel to 178 VGPRs = 2 waves per SIMD).
// Symmetric batches are virtual: direction b < Bh queries image A[b] against B[b], direction
// b >= Bh queries B[b-Bh] against A[b-Bh] (the reference concatenates the pyramids instead,
// model/network.py:213-222).
// two horizontally adjacent pixels of a map stored as FT, widened to fp32: an 8-byte (fp32, 4-byte aligned) or a 4-byte (fp16,
// 2-byte aligned) gather
struct __attribute__((packed, aligned(2))) h16x2u { _Float16 x, y; };
__device__ __forceinline__ f32x2u ld_pair(const float *q) { return *reinterpret_cast<const f32x2u *>(q); }
__device__ __forceinline__ f32x2u ld_pair(const _Float16 *q) {
    const h16x2u v = *reinterpret_cast<const h16x2u *>(q);
    f32x2u o;
    o.x = (float)v.x; o.y = (float)v.y;
    return o;
}

// symmetric batches: the two directions of one pair read the same two maps (query <-> support), so they are dispatched
// back to back (blockIdx.y -> direction) and the second reader finds the maps in the memory-side cache
__device__ __forceinline__ int ri_direction(int B, int Bh, unsigned by) {
    return Bh < B ? ((by & 1) ? (int)(by >> 1) + Bh : (int)(by >> 1)) : (int)by;
}

// Round 4, symmetric batches: a 1-D grid in XCD-BANDED order.  Blocks are dealt round-robin over the 8 XCDs (id & 7 = the XCD,
// id >> 3 = the block's place in that XCD's queue; a speed assumption only), and every map is read twice -- regularly by one
// direction of its pair (grid_feature), along the flow by the other (x_hat).  An XCD takes ONE band (an eighth of the cell blocks,
// consecutive rows) of BOTH directions of a pair, alternating between them, so that the band's rows of the two maps (a few MB:
// they stay in its 4 MB L2 for the band's duration) serve the second reader wherever the flow keeps it inside the band.  The
// per-direction order (blockIdx.y = direction, every XCD all over the map) fetched every map 2.2x through L2; measured on the
// bench's scene: 399 -> 332 us (G 320), 248 -> 211 (G 256), 121 -> 108 (G 128, with the plan).  Bands are floor(k q / 8) ..
// floor((k + 1) q / 8): when 8 does not divide q they differ by one block, every XCD gets ceil(q / 8) slots per direction (the
// spare slot of a short band returns at once) and the band an XCD takes rotates with the pair so that the long bands go round.
__device__ __forceinline__ unsigned ri_band_slots(unsigned q_blocks) { return (q_blocks + 7u) >> 3; }
__device__ __forceinline__ bool ri_banded(unsigned id, unsigned q_blocks, int Bh, int &b, unsigned &x) {
    const unsigned slots = ri_band_slots(q_blocks), xcd = id & 7u, s = id >> 3;
    const unsigned per_pair = 2u * slots;
    const unsigned p = s / per_pair, t = s - p * per_pair;
    const unsigned band = (xcd + p) & 7u;
    const unsigned lo = (band * q_blocks) >> 3, hi = ((band + 1u) * q_blocks) >> 3;
    x = lo + (t >> 1);
    b = (t & 1u) ? (int)p + Bh : (int)p;
    return x < hi;
}
inline bool ri_bands(int B, int Bh, unsigned q_blocks) { return Bh < B && q_blocks >= 8; }
inline unsigned ri_banded_blocks(int B, unsigned q_blocks) { return 8u * ((q_blocks + 7u) >> 3) * (unsigned)B; }

struct RiArgs {
    const void *fa, *fb;   // feature maps of the two images (Bh, C, Hs, Ws), fp32 or fp16
    const float *flow, *dw, *db;
    float *d;
    long d_bs;
    int B, Bh, C, Hs, Ws, G, Dd;
    float disp_scale;
};

// KEEP: the grid_feature planes (first C channels of d: grid_sample(x, cell centres), a function of x and the grid only) are
// already in d from an earlier call with the same x and G -- the second refiner iteration at a scale (num_itr = 2,
// model/network.py:257-268, calls the refiner again with a new flow): only x_hat and the displacement embedding are rewritten.
template <typename FT, bool KEEP = false>
__device__ __forceinline__ void refiner_input_cell(const RiArgs &args, int b, unsigned cell) {  // cell: the thread's index inside the direction
    const FT *__restrict__ fa = static_cast<const FT *>(args.fa);
    const FT *__restrict__ fb = static_cast<const FT *>(args.fb);
    const float *__restrict__ flow = args.flow, *__restrict__ dw = args.dw, *__restrict__ db = args.db;
    float *__restrict__ d = args.d;
    const long d_bs = args.d_bs;
    const int Bh = args.Bh, C = args.C, Hs = args.Hs, Ws = args.Ws, G = args.G, Dd = args.Dd;
    const float disp_scale = args.disp_scale;
    const float lo = (float)(-1 + 1.0 / G), hi = (float)(1 - 1.0 / G);
    const unsigned plane = (unsigned)(Hs * Ws), GG = (unsigned)(G * G);
    if (cell >= GG) return;
#ifndef GFN_RI_WAVE2D
#define GFN_RI_WAVE2D 1
#endif
    if (GFN_RI_WAVE2D && (G & 31) == 0) {
        // Round 6: a wave's 64 cells are 2 grid rows x 32 columns instead of 1 x 64.  The x_hat gathers follow the flow: under the
        // bench's homographies a 64-lane pair gather of a 1 x 64 wave touches 13.9 128-byte lines (rows drift with the rotation), of a
        // 2 x 32 wave 11.3; the regular grid_feature gather goes from 4 to 5 lines, the stores stay two full lines per instruction
        // (2 x 128 bytes).  A permutation of the cells inside a direction: results are unchanged.  (4 x 16 waves for grids that are a
        // multiple of 16 only -- G = 80, 48, 240 -- measured 0.5-0.9 % SLOWER: four rows of grid_feature gathers cost more than the x_hat gathers save.)
        const unsigned w = cell >> 6, l = cell & 63u, wpr = (unsigned)G >> 5;   // waves per row pair
        const unsigned rp = w / wpr, cb = w - rp * wpr;
        cell = (2u * rp + (l >> 5)) * (unsigned)G + cb * 32u + (l & 31u);
    }
    const int i = (int)(cell / (unsigned)G), j = (int)(cell - (unsigned)i * (unsigned)G);
    const FT *q = (b < Bh ? fa + (size_t)b * C * plane : fb + (size_t)(b - Bh) * C * plane);  // query map
    const FT *sm = (b < Bh ? fb + (size_t)b * C * plane : fa + (size_t)(b - Bh) * C * plane); // support map
    const float cx = gfn::linspace_at(lo, hi, G, j), cy = gfn::linspace_at(lo, hi, G, i);  // network.py:539-546
    const float *fl = flow + (size_t)b * 2 * GG;
    const float fx = fl[cell], fy = fl[GG + cell];
    float *o = d + (size_t)b * d_bs;
    if (Ws >= 2) {
        const BilinP sa = bilin_pairs(cx, cy, Ws, Hs);  // grid_feature = grid_sample(x, im_A_coords)   network.py:547
        const BilinP sb = bilin_pairs(fx, fy, Ws, Hs);  // x_hat = grid_sample(y, flow)                 network.py:537
        for (int c0 = 0; c0 < C; c0 += 8) {
            f32x2u va[8][2], vb[8][2];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const FT *qp = q + (size_t)min(c0 + k, C - 1) * plane, *sp = sm + (size_t)min(c0 + k, C - 1) * plane;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    if (!KEEP) va[k][e] = ld_pair(qp + sa.o[e]);
                    vb[k][e] = ld_pair(sp + sb.o[e]);
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (c0 + k < C) {
                    float ra = 0.f, rb = 0.f;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        if (!KEEP) {
                            ra += va[k][e].x * sa.w[2 * e];
                            ra += va[k][e].y * sa.w[2 * e + 1];
                        }
                        rb += vb[k][e].x * sb.w[2 * e];
                        rb += vb[k][e].y * sb.w[2 * e + 1];
                    }
                    if (!KEEP) (o + (size_t)(c0 + k) * GG)[cell] = ra;  // read back at once as the local correlation's f0: stays cached
                    __builtin_nontemporal_store(rb, o + (size_t)(C + c0 + k) * GG + cell);
                }
            }
        }
    } else {  // one-column maps: no pair to fetch
        const BilinC sa = bilin_clamped(cx, cy, Ws, Hs);
        const BilinC sb = bilin_clamped(fx, fy, Ws, Hs);
        for (int c = 0; c < C; ++c) {
            float ra = 0.f, rb = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (!KEEP) ra += (float)(q + (size_t)c * plane)[sa.o[e]] * sa.w[e];
                rb += (float)(sm + (size_t)c * plane)[sb.o[e]] * sb.w[e];
            }
            if (!KEEP) (o + (size_t)c * GG)[cell] = ra;
            (o + (size_t)(C + c) * GG)[cell] = rb;
        }
    }
    // disp_emb(40/32 * scale_factor * (flow - im_A_coords))                                  network.py:548-549
    const float dx = disp_scale * (fx - cx), dy = disp_scale * (fy - cy);
    for (int k = 0; k < Dd; ++k) __builtin_nontemporal_store(dw[k * 2 + 0] * dx + dw[k * 2 + 1] * dy + db[k], o + (size_t)(2 * C + k) * GG + cell);
}


}  // namespace gfn_ri
