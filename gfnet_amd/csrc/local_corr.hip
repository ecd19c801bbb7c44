// local_corr.hip -- fused local correlation for gfx950 (MI355X).
//
// Replaces utils/local_correlation.py:4-72 of the reference (called at model/network.py:553-554):
//   out[b,k,i,j] = sum_c f0[b,c,i,j]/sqrt(C) * grid_sample(f1[b,c], flow[b,:,i,j] + window[k])
// The reference materialises a (C,G,G,K) sampled tensor per batch element (26-59 MB) and makes
// three passes over it; here nothing but the (K,G,G) result is written.
//
// Fast path (one feature pixel per window tap, i.e. every call the reference makes):
//   * all K taps of a cell share one pair of bilinear fractions, so for each cell
//         D[y][x]   = sum_c f0[c] * f1[c][Y0+y][X0+x]        (y,x over the (2r+2)^2 patch)
//         out[ky,kx] = (w00 D[ky][kx] + w01 D[ky][kx+1] + w10 D[ky+1][kx] + w11 D[ky+1][kx+1])/sqrt(C)
//     which is 3x fewer FMAs and 4x fewer feature reads than sampling every tap;
//   * a workgroup (512 threads, 8 waves) owns a tile of 2*ROUNDS x 16 grid cells.  The bounding
//     box of the tile's patches is staged from f1 (NCHW, coalesced row reads, L2-resident) into
//     LDS as [pixel][16 channels + 4 pad] (80-byte slots), 16 channels at a time;
//   * D-stage: 16 lanes per cell, each lane owns patch positions p = s+16t.  A ds_read_b128 lane
//     group (the hardware's 16-lane b128 groups, MI355X_MICROARCH.md section LDS) reads 16
//     consecutive patch positions of ONE cell; with the 20-dword slot and a row pitch == patch
//     width (mod 16) these are 16 distinct 4-bank groups -> conflict-free 256 B/clk reads;
//     f0 is held in registers (16 per lane), accumulators in registers (<= 16 per round);
//   * epilogue: D goes through LDS (aliasing the stage), each wave combines 64 cells x taps and
//     stores 64-byte row segments of the (K,G,G) output.
//   * the reference adds the window offsets in normalised fp32 coordinates, tap by tap, so its
//     taps sit on whole-pixel steps only to ~3e-5 px (at W=280).  To stay bit-faithful the
//     epilogue uses, per cell and per tap row/column, the fraction of exactly that fp32 sequence
//     (a small LDS table); a cell for which some tap's floor() disagrees with (origin + tap
//     index) -- the centre within rounding of a pixel boundary, or a non-finite flow -- is redone
//     by the general per-tap routine at the end of the launch (about one cell in 10^4);
//   * tiles whose bounding box does not fit the stage (wild flow) are staged per 2x16 round, and
//     rounds that still do not fit use the general per-tap routine inside the same launch.
// General path (any C, radius, non-integer tap spacing: grid_based_correlation, pooled levels):
//   one thread per (cell, tap), per-tap bilinear gather exactly as grid_sample does it.
//
// Block->tile mapping is XCD-aware (common.h): consecutive tiles of one image stay on one XCD so
// the 3-4x halo re-reads of f1 are served by that XCD's L2, not HBM.
#include "common.h"

namespace {

constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;
constexpr int kChunk = 16;                // channels staged per pass
constexpr int kSlotV4 = kChunk / 4 + 1;   // float4s per staged pixel: 4 data + 1 pad = 80 B
constexpr int kStageBytes = 72 * 1024;    // stage buffer (aliased by the D buffer in the epilogue)
constexpr int kCapSlots = kStageBytes / (kSlotV4 * 16) - 1;  // pixels that fit, minus the zero slot
static_assert((kCapSlots + 1) * kSlotV4 < 65536, "stage indices are packed in 16 bits");
constexpr int kTileW = 16;
constexpr int kFar = 1 << 28;             // patch origin of a cell that samples nothing

struct LcParams {
    const float *f0;
    const float *f1;
    const float *flow;
    float *out;
    long f0_bs, out_bs;
    int B, C, G, H, W;
    int tiles_x, tiles_y;
    float sqrt_c;
    // general path only
    int r, win_h, win_w, grid_based;
#ifdef GFN_ABLATE
    int dbg;  // timing experiments only (tools/probe_local_corr.py): bit mask of stages to skip
#endif
};

#ifdef GFN_ABLATE
#define ABL(p, bit) (((p).dbg & (bit)) != 0)
#else
#define ABL(p, bit) false
#endif

struct Region {
    int x0, y0, w, h, pitch;
};

// lane -> (b128 hardware lane group, index inside the group).  ds_read_b128 is serviced in four
// 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, and the same +32.
__device__ __forceinline__ void lane_group(int lane, int &g, int &s) {
    const int m = lane & 31;
    const bool even = (m < 4) | ((m >= 12) & (m < 16)) | ((m >= 20) & (m < 28));
    if (even)
        s = (m < 4) ? m : (m < 16 ? m - 8 : m - 12);
    else
        s = (m < 12) ? m - 4 : (m < 20 ? m - 8 : m - 16);
    g = ((lane >> 5) << 1) | (even ? 0 : 1);
}

// Normalised -> pixel coordinate exactly as grid_sample(align_corners=False) un-normalises.
__device__ __forceinline__ float unnorm(float g, int size) { return ((g + 1.f) * (float)size - 1.f) / 2.f; }

// ---- general per-tap evaluation (mirrors the reference op for op) ---------------------------
__device__ float tap_general(const LcParams &p, int b, int i, int j, int ky, int kx, int D, float nx, float ny) {
    float ylo, yhi, xlo, xhi;
    if (p.grid_based) {
        ylo = (float)(-2.0 * p.r / p.G); yhi = (float)(2.0 * p.r / p.G);
        xlo = ylo; xhi = yhi;
    } else {
        ylo = (float)(-2.0 * p.r / p.win_h); yhi = (float)(2.0 * p.r / p.win_h);
        xlo = (float)(-2.0 * p.r / p.win_w); xhi = (float)(2.0 * p.r / p.win_w);
    }
    const float gx = nx + gfn::linspace_at(xlo, xhi, D, kx);
    const float gy = ny + gfn::linspace_at(ylo, yhi, D, ky);
    const float ix = unnorm(gx, p.W), iy = unnorm(gy, p.H);
    float fx = floorf(ix), fy = floorf(iy);
    const bool sane = (fx > -1e6f) & (fx < 1e6f) & (fy > -1e6f) & (fy < 1e6f);
    const int x0 = sane ? (int)fx : -4, y0 = sane ? (int)fy : -4;
    const float w00 = (fx + 1.f - ix) * (fy + 1.f - iy), w01 = (ix - fx) * (fy + 1.f - iy);
    const float w10 = (fx + 1.f - ix) * (iy - fy), w11 = (ix - fx) * (iy - fy);
    const bool xa = (unsigned)x0 < (unsigned)p.W, xb = (unsigned)(x0 + 1) < (unsigned)p.W;
    const bool ya = (unsigned)y0 < (unsigned)p.H, yb = (unsigned)(y0 + 1) < (unsigned)p.H;
    const float *f0p = p.f0 + (size_t)b * p.f0_bs + (size_t)i * p.G + j;
    const float *f1p = p.f1 + (size_t)b * p.C * p.H * p.W;
    const size_t plane = (size_t)p.H * p.W, cs = (size_t)p.G * p.G;
    const long o00 = (long)y0 * p.W + x0;
    float acc = 0.f;
    for (int c = 0; c < p.C; ++c) {
        const float *pl = f1p + c * plane;
        float s = 0.f;
        if (ya & xa) s += pl[o00] * w00;
        if (ya & xb) s += pl[o00 + 1] * w01;
        if (yb & xa) s += pl[o00 + p.W] * w10;
        if (yb & xb) s += pl[o00 + p.W + 1] * w11;
        acc += (f0p[c * cs] / p.sqrt_c) * s;
    }
    return acc;
}

__device__ __forceinline__ void cell_coords(const LcParams &p, int b, int i, int j, float &nx, float &ny) {
    if (p.flow) {
        nx = p.flow[(((size_t)b * 2 + 0) * p.G + i) * p.G + j];
        ny = p.flow[(((size_t)b * 2 + 1) * p.G + i) * p.G + j];
    } else {  // identity grid (local_correlation.py:21-30)
        nx = gfn::linspace_at((float)(-1 + 1.0 / p.win_w), (float)(1 - 1.0 / p.win_w), p.win_w, j);
        ny = gfn::linspace_at((float)(-1 + 1.0 / p.win_h), (float)(1 - 1.0 / p.win_h), p.win_h, i);
    }
}

__global__ __launch_bounds__(256) void local_corr_general_kernel(LcParams p) {
    const int D = 2 * p.r + 1, K = D * D;
    const long total = (long)p.B * K * p.G * p.G;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int j = (int)(idx % p.G);
        long t = idx / p.G;
        const int i = (int)(t % p.G);
        t /= p.G;
        const int k = (int)(t % K);
        const int b = (int)(t / K);
        float nx, ny;
        cell_coords(p, b, i, j, nx, ny);
        p.out[(size_t)b * p.out_bs + ((size_t)k * p.G + i) * p.G + j] = tap_general(p, b, i, j, k / D, k % D, D, nx, ny);
    }
}

// ---- fast tiled kernel -----------------------------------------------------------------------
template <int UN>  // wave-iterations in flight: all their loads are issued before the first LDS write
__device__ __forceinline__ void stage_region(float4 *s4, const float *f1c, int H, int W, const Region &rg, int wave,
                                             int lane) {
    const int npx = rg.w * rg.h;
    const int nwi = ((npx + 63) >> 6) * 4;  // wave-iterations: 4 channel groups x runs of 64 pixels
    const float inv_w = 1.0f / (float)rg.w;
    const size_t plane = (size_t)H * W;
    for (int wi0 = wave; wi0 < nwi; wi0 += kWaves * UN) {
        float4 v[UN];
        int dst[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int wi = wi0 + u * kWaves;
            const int cg = wi & 3;
            const int q = ((wi >> 2) << 6) + lane;
            dst[u] = -1;
            if (wi < nwi && q < npx) {
                const int y = (int)(((float)q + 0.5f) * inv_w);  // exact for q < 2^16, w < 2^10
                const int x = q - y * rg.w;
                const float *src = f1c + (size_t)(cg * 4) * plane + (size_t)(rg.y0 + y) * W + (rg.x0 + x);
                v[u].x = src[0];
                v[u].y = src[plane];
                v[u].z = src[2 * plane];
                v[u].w = src[3 * plane];
                dst[u] = (y * rg.pitch + x) * kSlotV4 + cg;
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
            if (dst[u] >= 0) s4[dst[u]] = v[u];
    }
}

template <int R, int ROUNDS>
__global__ __launch_bounds__(kThreads, 4) void local_corr_tile_kernel(LcParams p) {
    constexpr int PW = 2 * R + 2;            // patch width: taps -R..R plus the +1 bilinear neighbour
    constexpr int P = PW * PW;               // patch positions per cell
    constexpr int NP = (P + 15) / 16;        // positions per lane
    constexpr int D = 2 * R + 1, K = D * D;
    constexpr int TH = 2 * ROUNDS;           // tile height in cells
    constexpr int NC = 32 * ROUNDS;          // cells per tile
    constexpr int DS = P + 1;                // D-buffer cell stride (odd: conflict-free epilogue reads)
    constexpr int TS = 2 * D + 1;            // fraction-table cell stride (odd)
    static_assert((NC * DS + NC * TS) * 4 <= kStageBytes, "D buffer + fraction table must fit in the stage they alias");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *s4 = reinterpret_cast<float4 *>(smem);
    float *dbuf = reinterpret_cast<float *>(smem);
    int *cellX0 = reinterpret_cast<int *>(smem + kStageBytes);
    int *cellY0 = cellX0 + NC;
    float *cellNx = reinterpret_cast<float *>(cellY0 + NC);  // normalised centre (flow) of the cell
    float *cellNy = cellNx + NC;
    int *bbox = reinterpret_cast<int *>(cellNy + NC);  // [ROUNDS][4] = x0,y0,x1,y1
    int *cellSlow = bbox + ROUNDS * 4;                  // [NC] 1 = redo this cell with the per-tap routine
    int *nSlow = cellSlow + NC;                         // number of such cells in the tile
    float *tab = dbuf + NC * DS;                        // [NC][TS] per-tap fractions (aliases the stage)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned wid = gfn::xcd_remap(blockIdx.x, gridDim.x);
    const int tiles = p.tiles_x * p.tiles_y;
    const int b = wid / tiles;
    const int tile = wid - b * tiles;
    const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    const int G = p.G, H = p.H, W = p.W;

    // ---- per-cell setup: pixel coordinate, patch origin, bounding boxes ----------------------
    if (tid < ROUNDS * 4) bbox[tid] = (tid & 2) ? -kFar : kFar;
    if (tid == 0) *nSlow = 0;
    __syncthreads();
    const float xlo = (float)(-2.0 * R / W), xhi = (float)(2.0 * R / W);
    const float ylo = (float)(-2.0 * R / H), yhi = (float)(2.0 * R / H);
    if (tid < NC) {
        const int ci = tid >> 4, cj = tid & 15;
        const int gi = ty * TH + ci, gj = tx * kTileW + cj;
        int X0 = kFar, Y0 = kFar, slow = 0;
        float nx = 0.f, ny = 0.f;
        if (gi < G && gj < G) {
            cell_coords(p, b, gi, gj, nx, ny);
            // patch origin = floor of the reference's own fp32 coordinate of tap 0, minus nothing:
            // taps kx=0..2R then read columns kx and kx+1 of the patch
            const float fx = floorf(unnorm(nx + gfn::linspace_at(xlo, xhi, D, 0), W));
            const float fy = floorf(unnorm(ny + gfn::linspace_at(ylo, yhi, D, 0), H));
            if ((fx > -1e6f) & (fx < 1e6f) & (fy > -1e6f) & (fy < 1e6f)) {  // false for nan/inf
                X0 = (int)fx;
                Y0 = (int)fy;
                const int x0 = max(X0, 0), x1 = min(X0 + PW, W), y0 = max(Y0, 0), y1 = min(Y0 + PW, H);
                if (x0 < x1 && y0 < y1) {
                    int *bb = bbox + (tid >> 5) * 4;
                    atomicMin(bb + 0, x0);
                    atomicMin(bb + 1, y0);
                    atomicMax(bb + 2, x1);
                    atomicMax(bb + 3, y1);
                }
            } else {
                slow = 1;  // non-finite / absurd flow: let the per-tap routine decide
                atomicAdd(nSlow, 1);
            }
        }
        cellX0[tid] = X0;
        cellY0[tid] = Y0;
        cellNx[tid] = nx;
        cellNy[tid] = ny;
        cellSlow[tid] = slow;
    }
    // the zero slot (index kCapSlots) is what every out-of-image tap reads
    if (tid < kSlotV4) s4[kCapSlots * kSlotV4 + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    // ---- choose the staging regions (block-uniform) -------------------------------------------
    Region reg[ROUNDS];
    bool fit[ROUNDS];
    bool whole;
    {
        int ux0 = kFar, uy0 = kFar, ux1 = -kFar, uy1 = -kFar;
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            const int x0 = bbox[rd * 4 + 0], y0 = bbox[rd * 4 + 1], x1 = bbox[rd * 4 + 2], y1 = bbox[rd * 4 + 3];
            ux0 = min(ux0, x0); uy0 = min(uy0, y0); ux1 = max(ux1, x1); uy1 = max(uy1, y1);
            Region r;
            r.x0 = x0; r.y0 = y0;
            r.w = max(x1 - x0, 0); r.h = max(y1 - y0, 0);
            r.pitch = r.w + ((PW - r.w) & 15);
            reg[rd] = r;
            fit[rd] = (long)r.pitch * r.h <= kCapSlots;
        }
        Region u;
        u.x0 = ux0; u.y0 = uy0;
        u.w = max(ux1 - ux0, 0); u.h = max(uy1 - uy0, 0);
        u.pitch = u.w + ((PW - u.w) & 15);
        whole = (long)u.pitch * u.h <= kCapSlots;
        if (whole) {
#pragma unroll
            for (int rd = 0; rd < ROUNDS; ++rd) { reg[rd] = u; fit[rd] = true; }
        }
    }

    // ---- per-lane D-stage addressing -------------------------------------------------------
    int g, s16;
    lane_group(lane, g, s16);
    const int cr = wave * 4 + g;  // cell inside a round (0..31): row cr>>4, column cr&15
    unsigned apk[ROUNDS][(NP + 1) / 2];  // float4 index (< 2^16) of each (round, pass) patch pixel, two per register
    float acc[ROUNDS][NP];
    const float *f0c[ROUNDS];     // this lane's f0 column (channel stride G*G), or null
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
        const int cell = rd * 32 + cr;
        const int X0 = cellX0[cell], Y0 = cellY0[cell];
        const int gi = ty * TH + (cell >> 4), gj = tx * kTileW + (cell & 15);
        f0c[rd] = (gi < G && gj < G) ? p.f0 + (size_t)b * p.f0_bs + (size_t)gi * G + gj : nullptr;
#pragma unroll
        for (int t = 0; t < NP; ++t) {
            const int pp = s16 + 16 * t;
            const int yy = pp / PW, xx = pp - yy * PW;
            const int X = X0 + xx, Y = Y0 + yy;
            const bool in = (pp < P) & ((unsigned)X < (unsigned)W) & ((unsigned)Y < (unsigned)H);
            const int slot = in ? (Y - reg[rd].y0) * reg[rd].pitch + (X - reg[rd].x0) : kCapSlots;
            const unsigned a = (unsigned)(slot * kSlotV4);
            if (t & 1)
                apk[rd][t >> 1] |= a << 16;
            else
                apk[rd][t >> 1] = a;
            acc[rd][t] = 0.f;
        }
    }

    // ---- main loop: 16 channels at a time ----------------------------------------------------
    const size_t cs = (size_t)G * G;
    const float *f1b = p.f1 + (size_t)b * p.C * H * W;
    constexpr int UN = (ROUNDS * NP >= 32) ? 1 : ((ROUNDS * NP >= 24 || ROUNDS > 2) ? 2 : 4);
    for (int c0 = 0; c0 < p.C; c0 += kChunk) {
        // keep the packed indices packed: without this the unpacking is hoisted out of the loop and
        // the unpacked copies cost NP more registers per round (spills at r = 6, 7)
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd)
#pragma unroll
            for (int h = 0; h < (NP + 1) / 2; ++h) asm volatile("" : "+v"(apk[rd][h]));
        if (whole) {  // the usual case: one stage per channel chunk serves every round
            __syncthreads();  // everyone is done reading the previous contents
            if (!ABL(p, 1)) stage_region<UN>(s4, f1b + (size_t)c0 * H * W, H, W, reg[0], wave, lane);
            __syncthreads();
        }
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            if (!fit[rd]) continue;
            float f[kChunk];
#pragma unroll
            for (int k = 0; k < kChunk; ++k) f[k] = (f0c[rd] && !ABL(p, 4)) ? f0c[rd][(size_t)(c0 + k) * cs] : 0.f;
            if (!whole) {
                __syncthreads();
                if (!ABL(p, 1)) stage_region<1>(s4, f1b + (size_t)c0 * H * W, H, W, reg[rd], wave, lane);
                __syncthreads();
            }
            if (ABL(p, 2)) continue;
#pragma unroll
            for (int t = 0; t < NP; ++t) {
                const float4 *q = s4 + ((t & 1) ? (apk[rd][t >> 1] >> 16) : (apk[rd][t >> 1] & 0xFFFFu));
                const float4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3];
                float a = acc[rd][t];
                a = fmaf(f[0], v0.x, a);  a = fmaf(f[1], v0.y, a);  a = fmaf(f[2], v0.z, a);  a = fmaf(f[3], v0.w, a);
                a = fmaf(f[4], v1.x, a);  a = fmaf(f[5], v1.y, a);  a = fmaf(f[6], v1.z, a);  a = fmaf(f[7], v1.w, a);
                a = fmaf(f[8], v2.x, a);  a = fmaf(f[9], v2.y, a);  a = fmaf(f[10], v2.z, a); a = fmaf(f[11], v2.w, a);
                a = fmaf(f[12], v3.x, a); a = fmaf(f[13], v3.y, a); a = fmaf(f[14], v3.z, a); a = fmaf(f[15], v3.w, a);
                acc[rd][t] = a;
            }
        }
    }

    // ---- epilogue: D -> LDS, per-tap fractions, bilinear combination, coalesced stores -------
    __syncthreads();
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
        const int cell = rd * 32 + cr;
#pragma unroll
        for (int t = 0; t < NP; ++t) {
            const int pp = s16 + 16 * t;
            if (pp < P) dbuf[cell * DS + pp] = acc[rd][t];
        }
    }
    // fraction table: the reference's fp32 coordinate of every tap column / row of every cell
    // (local_correlation.py:55 adds window offsets in normalised units, grid_sample un-normalises)
    for (int e = tid; e < NC * 2 * D && !ABL(p, 32); e += kThreads) {
        const int cell = e / (2 * D), a = e - cell * (2 * D);
        const bool isy = a >= D;
        const int k = isy ? a - D : a;
        const float n = isy ? cellNy[cell] : cellNx[cell];
        const float pix = unnorm(n + (isy ? gfn::linspace_at(ylo, yhi, D, k) : gfn::linspace_at(xlo, xhi, D, k)),
                                 isy ? H : W);
        const float fl = floorf(pix);
        const int origin = isy ? cellY0[cell] : cellX0[cell];
        // tap k must start at patch column/row k; if rounding moved its floor(), redo the cell per tap
        if (origin != kFar && !(fl == (float)(origin + k))) {
            cellSlow[cell] = 1;
            atomicAdd(nSlow, 1);
        }
        tab[cell * TS + a] = pix - fl;
    }
    __syncthreads();
    {
        constexpr int NCB = NC / 64;      // 64-cell blocks per tile
        constexpr int WPB = kWaves / NCB;  // waves sharing one block of cells
        const int cell = (wave / WPB) * 64 + lane;
        const int rd = cell >> 5;
        const int gi = ty * TH + (cell >> 4), gj = tx * kTileW + (cell & 15);
        bool fast = false;
#pragma unroll
        for (int q = 0; q < ROUNDS; ++q) fast |= (q == rd) & fit[q];
        if (fast && gi < G && gj < G && !cellSlow[cell] && !ABL(p, 8)) {
            const float *dc = dbuf + cell * DS;
            const float *tc = tab + cell * TS;
            float *o = p.out + (size_t)b * p.out_bs + (size_t)gi * G + gj;
            for (int k = wave % WPB; k < K; k += WPB) {
                const int ky = k / D, kx = k - ky * D;
                const float wx1 = tc[kx], wy1 = tc[D + ky];
                const float wx0 = 1.f - wx1, wy0 = 1.f - wy1;
                const float *d = dc + ky * PW + kx;
                // corner order and weights as grid_sample: nw, ne, sw, se
                float v = d[0] * (wx0 * wy0);
                v += d[1] * (wx1 * wy0);
                v += d[PW] * (wx0 * wy1);
                v += d[PW + 1] * (wx1 * wy1);
                o[(size_t)k * cs] = v / p.sqrt_c;
            }
        }
    }

    // ---- what the tiled path could not do: whole rounds whose windows did not fit the stage, and
    //      single cells flagged above.  General per-tap routine, same launch. ----------------------
    bool any_slow = *nSlow != 0;
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) any_slow |= !fit[rd];
    if (any_slow) {  // block-uniform, rare
        LcParams q = p;
        q.r = R; q.win_h = H; q.win_w = W; q.grid_based = 0;
        for (int cell = 0; cell < NC && !ABL(p, 16); ++cell) {
            bool slow = cellSlow[cell] != 0;
#pragma unroll
            for (int rd = 0; rd < ROUNDS; ++rd) slow |= ((cell >> 5) == rd) & !fit[rd];
            if (!slow) continue;  // block-uniform
            const int gi = ty * TH + (cell >> 4), gj = tx * kTileW + (cell & 15);
            if (gi >= G || gj >= G) continue;
            for (int k = tid; k < K; k += kThreads)
                p.out[(size_t)b * p.out_bs + ((size_t)k * G + gi) * G + gj] =
                    tap_general(q, b, gi, gj, k / D, k % D, D, cellNx[cell], cellNy[cell]);
        }
    }
}

template <int R, int ROUNDS>
int launch_tile(const LcParams &p0, hipStream_t stream) {
    LcParams p = p0;
    constexpr int NC = 32 * ROUNDS;
    p.tiles_x = (p.G + kTileW - 1) / kTileW;
    p.tiles_y = (p.G + 2 * ROUNDS - 1) / (2 * ROUNDS);
    const size_t lds = kStageBytes + NC * 20 + ROUNDS * 16 + 16;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(local_corr_tile_kernel<R, ROUNDS>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const unsigned grid = (unsigned)p.B * p.tiles_x * p.tiles_y;
    hipLaunchKernelGGL((local_corr_tile_kernel<R, ROUNDS>), dim3(grid), dim3(kThreads), lds, stream, p);
    return gfn::check_launch("local_corr_tile_kernel");
}

}  // namespace

GFN_EXPORT int gfn_local_corr_fwd_ex(const float *f0, int64_t f0_bs, const float *f1, const float *flow, float *out,
                                     int64_t out_bs, int B, int C, int G, int H, int W, int r, int grid_based,
                                     int win_h, int win_w, int variant, gfn_stream_t stream) {
    if (!f0 || !f1 || !out) return gfn::fail(GFN_ERR_INVALID_ARG, "local_corr: null tensor pointer");
    if (B < 0 || C <= 0 || G <= 0 || H <= 0 || W <= 0 || r < 0 || win_h <= 0 || win_w <= 0)
        return gfn::fail(GFN_ERR_INVALID_ARG, "local_corr: bad size B=%d C=%d G=%d H=%d W=%d r=%d", B, C, G, H, W, r);
    const long K = (long)(2 * r + 1) * (2 * r + 1);
    if (f0_bs < (long)C * G * G || out_bs < K * G * G)
        return gfn::fail(GFN_ERR_INVALID_ARG, "local_corr: batch stride smaller than one batch element");
    if (!flow && !(G == win_h && G == win_w))
        return gfn::fail(GFN_ERR_INVALID_ARG, "local_corr: flow=NULL needs num_grid == h == w (got G=%d h=%d w=%d)", G,
                         win_h, win_w);
    if ((long)B * K * G * G >= (1L << 40) || (long)C * H * W >= (1L << 31))
        return gfn::fail(GFN_ERR_INVALID_ARG, "local_corr: tensor too large");
    if (B == 0) return GFN_OK;
    hipStream_t s = (hipStream_t)stream;
    LcParams p;
    p.f0 = f0; p.f1 = f1; p.flow = flow; p.out = out;
    p.f0_bs = f0_bs; p.out_bs = out_bs;
    p.B = B; p.C = C; p.G = G; p.H = H; p.W = W;
    p.tiles_x = p.tiles_y = 0;
    p.sqrt_c = (float)sqrt((double)C);
    p.r = r; p.win_h = win_h; p.win_w = win_w; p.grid_based = grid_based;
#ifdef GFN_ABLATE
    p.dbg = variant >> 8;
    variant &= 0xff;
#endif

    const bool fast_ok = variant == 0 && !grid_based && win_h == H && win_w == W && (C % kChunk) == 0 && r >= 1 && r <= 7;
    if (fast_ok) {
        switch (r) {
            case 1: return launch_tile<1, 4>(p, s);
            case 2: return launch_tile<2, 4>(p, s);
            case 3: return launch_tile<3, 2>(p, s);
            case 4: return launch_tile<4, 2>(p, s);
            case 5: return launch_tile<5, 2>(p, s);
            case 6: return launch_tile<6, 2>(p, s);
            case 7: return launch_tile<7, 2>(p, s);
        }
    }
    const long total = (long)B * K * G * G;
    const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(local_corr_general_kernel, dim3(grid), dim3(256), 0, s, p);
    return gfn::check_launch("local_corr_general_kernel");
}

GFN_EXPORT int gfn_local_corr_fwd(const float *f0, int64_t f0_bs, const float *f1, const float *flow, float *out,
                                  int64_t out_bs, int B, int C, int G, int H, int W, int r, int grid_based, int win_h,
                                  int win_w, gfn_stream_t stream) {
    return gfn_local_corr_fwd_ex(f0, f0_bs, f1, flow, out, out_bs, B, C, G, H, W, r, grid_based, win_h, win_w, 0,
                                 stream);
}

namespace {
__global__ __launch_bounds__(256) void avg_pool2_kernel(const float *in, float *out, int BC, int H, int W) {
    const int Ho = H / 2, Wo = W / 2;
    const long total = (long)BC * Ho * Wo;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % Wo);
        const long t = idx / Wo;
        const int y = (int)(t % Ho);
        const long pl = t / Ho;
        const float *s = in + (pl * H + 2 * y) * W + 2 * x;
        out[idx] = (s[0] + s[1] + s[W] + s[W + 1]) / 4.f;
    }
}
}  // namespace

GFN_EXPORT int gfn_avg_pool2(const float *in, float *out, int BC, int H, int W, gfn_stream_t stream) {
    if (!in || !out || BC < 0 || H < 2 || W < 2) return gfn::fail(GFN_ERR_INVALID_ARG, "avg_pool2: bad argument");
    const long total = (long)BC * (H / 2) * (W / 2);
    if (total == 0) return GFN_OK;
    const unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(avg_pool2_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, out, BC, H, W);
    return gfn::check_launch("avg_pool2_kernel");
}
