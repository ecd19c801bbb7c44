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
#include "refiner_input.h"

namespace {

constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;
constexpr int kChunk = 16;                // channels staged per pass
constexpr int kSlotV4 = kChunk / 4 + 1;   // float4s per staged pixel: 4 data + 1 pad = 80 B
constexpr int kStageBytes = 68 * 1024;    // stage buffer (aliased by the D buffer in the epilogue)
constexpr int kCapSlots = kStageBytes / (kSlotV4 * 16) - 1;  // pixels that fit, minus the zero slot
static_assert((kCapSlots + 1) * kSlotV4 < 65536, "stage indices are packed in 16 bits");
constexpr int kTileW = 16;
constexpr int kMaxLds = 150 * 1024;       // dynamic LDS a tiled launch may ask for
constexpr int kFar = 1 << 28;             // patch origin of a cell that samples nothing
constexpr int kTodoHdr = 8;               // ints in front of the tile list in scratch: [0] tiles left to the second launch, [1] its queue head,
                                          // [2] its finished workgroups, [3] last call's [0], [4] cells redone per tap (running call),
                                          // [5] last call's [4], [6] tiles staged in halves (running call), [7] last call's [6]; [0..2], [4] and [6] are zero between calls

struct LcParams {
    const float *f0;
    const void *f1;          // feature maps (B or Bh, C, H, W), fp32 or fp16 (f16 != 0): BASELINE config 5 stores the pyramids in fp16
    const void *f1_second;   // symmetric batches: f1 of directions b >= Bh (NULL: f1 holds all B maps)
    int f16;
    int Bh;
    const float *flow;
    float *out;
    long f0_bs, out_bs;
    int B, C, G, H, W;
    int tiles_x, tiles_y;
    float sqrt_c, inv_sqrt_c;
    int r, win_h, win_w, grid_based;  // general path / flagged cells
    float win_xhi, win_yhi;           // tiled path: linspace end points 2r/W, 2r/H rounded to fp32
    float win_xstep, win_ystep;       // ... and the linspace steps (hi - lo) / (2r), fp32 division done on the host
    int *todo;                        // [kTodoHdr + B*tiles]: header (see kTodoHdr), then the ids of the tiles left to the second launch
    long todo_ints;
    int planned;                      // lean path: the plan is already in scratch (gfn_refiner_input_plan_fwd_dt wrote it)
    int mq;                           // r >= 5: the first launch is the matrix-core tile kernel (local_corr_mq.h)
    int *plan;                        // lean path: [4 * B*tiles] per-tile staging regions written by the plan launch (16-byte aligned)
#ifdef GFN_ABLATE
    int dbg;  // timing experiments only (tools/probe_local_corr.py): bit mask of stages to skip
#endif
};

#ifdef GFN_ABLATE
#define ABL(p, bit) (((p).dbg & (bit)) != 0)
// phase time stamps of one workgroup (tools/ablate_local_corr.py --stamps): s_memtime at the phase boundaries
#define STAMP(i) do { if (stamping) stamp[i] = __builtin_readcyclecounter(); } while (0)
#else
#define ABL(p, bit) false
#define STAMP(i) do { } while (0)
#endif

// f1 map of direction b.  Symmetric batches are virtual: the second half of the directions reads
// the other image's features (f1_second) instead of a concatenated copy (model/network.py:213-222).
template <typename FT>
__device__ __forceinline__ const FT *f1_of(const LcParams &p, int b) {
    const size_t chw = (size_t)p.C * p.H * p.W;
    return (b < p.Bh) ? static_cast<const FT *>(p.f1) + (size_t)b * chw : static_cast<const FT *>(p.f1_second) + (size_t)(b - p.Bh) * chw;
}
// a feature value as fp32 (fp16 storage is widened in registers; every sum stays fp32)
__device__ __forceinline__ float ldf(const float *q) { return *q; }
__device__ __forceinline__ float ldf(const _Float16 *q) { return (float)*q; }

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

// min over the 64 lanes of a wave, returned to every lane (scalar): four row shifts inside each row of 16, then the two
// row broadcasts of the GFX9 DPP set; lanes without a source keep their own value (min is idempotent).
__device__ __forceinline__ int wave_min_i32(int v) {
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));  // row_shr:1
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));  // row_shr:2
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));  // row_shr:4
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));  // row_shr:8
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));  // row_bcast:15 -> rows 1, 3
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));  // row_bcast:31 -> rows 2, 3
    return __builtin_amdgcn_readlane(v, 63);
}

// Normalised -> pixel coordinate exactly as grid_sample(align_corners=False) un-normalises.
__device__ __forceinline__ float unnorm(float g, int size) { return ((g + 1.f) * (float)size - 1.f) / 2.f; }

// ---- general per-tap evaluation (mirrors the reference op for op) ---------------------------
template <typename FT>
__device__ __forceinline__ float tap_general(const LcParams &p, int b, int i, int j, int ky, int kx, int D, float nx, float ny) {
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
    const FT *f1p = f1_of<FT>(p, b);
    const size_t plane = (size_t)p.H * p.W, cs = (size_t)p.G * p.G;
    const long o00 = (long)y0 * p.W + x0;
    // zero padding without branches: a corner outside the image reads pixel 0 with weight 0 (adds an exact 0), so the
    // gathers of 8 channels can all be in flight at once -- the branchy form exposed one L2 round trip per channel
    const long oa = (ya & xa) ? o00 : 0, ob = (ya & xb) ? o00 + 1 : 0, oc = (yb & xa) ? o00 + p.W : 0, od = (yb & xb) ? o00 + p.W + 1 : 0;
    const float wa = (ya & xa) ? w00 : 0.f, wb = (ya & xb) ? w01 : 0.f, wc = (yb & xa) ? w10 : 0.f, wd = (yb & xb) ? w11 : 0.f;
    float acc = 0.f;
    constexpr int UC = 8;
    for (int c0 = 0; c0 < p.C; c0 += UC) {
        float va[UC], vb[UC], vc[UC], vd[UC], q[UC];
#pragma unroll
        for (int u = 0; u < UC; ++u) {
            const int c = min(c0 + u, p.C - 1);
            const FT *pl = f1p + c * plane;
            va[u] = ldf(pl + oa); vb[u] = ldf(pl + ob); vc[u] = ldf(pl + oc); vd[u] = ldf(pl + od);
            q[u] = f0p[c * cs];
        }
#pragma unroll
        for (int u = 0; u < UC; ++u) {
            if (c0 + u < p.C) {
                float s = 0.f;
                s += va[u] * wa;
                s += vb[u] * wb;
                s += vc[u] * wc;
                s += vd[u] * wd;
                acc += (q[u] / p.sqrt_c) * s;
            }
        }
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

template <typename FT>
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
        p.out[(size_t)b * p.out_bs + ((size_t)k * p.G + i) * p.G + j] = tap_general<FT>(p, b, i, j, k / D, k % D, D, nx, ny);
    }
}

// ---- backward w.r.t. feature0 ---------------------------------------------------
// out[b,k,i,j] = sum_c (f0[b,c,i,j] / sqrt(c)) * S_c(k) with S_c(k) the bilinear sample of f1[b,c] at tap k; the reference
// lets gradients reach feature0 only (local_correlation.py:54-60: the sampling runs under no_grad).  Hence
//   grad_f0[b,c,i,j] = (sum_k grad_out[b,k,i,j] * S_c(k)) / sqrt(c):
// one thread per (cell, 8-channel group) walks the K taps with the forward's own coordinate arithmetic.
constexpr int kBwdCh = 8;
__global__ __launch_bounds__(256) void local_corr_bwd_f0_kernel(LcParams p, const float *__restrict__ gout, long gout_bs,
                                                                float *__restrict__ gf0, long gf0_bs) {
    const int D = 2 * p.r + 1, K = D * D;
    const int groups = (p.C + kBwdCh - 1) / kBwdCh;
    const long total = (long)p.B * groups * p.G * p.G;
    float ylo, yhi, xlo, xhi;
    if (p.grid_based) {
        ylo = (float)(-2.0 * p.r / p.G); yhi = (float)(2.0 * p.r / p.G);
        xlo = ylo; xhi = yhi;
    } else {
        ylo = (float)(-2.0 * p.r / p.win_h); yhi = (float)(2.0 * p.r / p.win_h);
        xlo = (float)(-2.0 * p.r / p.win_w); xhi = (float)(2.0 * p.r / p.win_w);
    }
    const size_t plane = (size_t)p.H * p.W, cs = (size_t)p.G * p.G;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int j = (int)(idx % p.G);
        long t = idx / p.G;
        const int i = (int)(t % p.G);
        t /= p.G;
        const int cg = (int)(t % groups);
        const int b = (int)(t / groups);
        const int c0 = cg * kBwdCh, nc = min(kBwdCh, p.C - c0);
        float nx, ny;
        cell_coords(p, b, i, j, nx, ny);
        const float *f1p = f1_of<float>(p, b) + (size_t)c0 * plane;
        const float *g = gout + (size_t)b * gout_bs + (size_t)i * p.G + j;
        float acc[kBwdCh];
#pragma unroll
        for (int c = 0; c < kBwdCh; ++c) acc[c] = 0.f;
        for (int k = 0; k < K; ++k) {
            const int ky = k / D, kx = k - ky * D;
            const float gx = nx + gfn::linspace_at(xlo, xhi, D, kx);
            const float gy = ny + gfn::linspace_at(ylo, yhi, D, ky);
            const float ix = unnorm(gx, p.W), iy = unnorm(gy, p.H);
            const float fx = floorf(ix), fy = floorf(iy);
            const bool sane = (fx > -1e6f) & (fx < 1e6f) & (fy > -1e6f) & (fy < 1e6f);
            const int x0 = sane ? (int)fx : -4, y0 = sane ? (int)fy : -4;
            const float w00 = (fx + 1.f - ix) * (fy + 1.f - iy), w01 = (ix - fx) * (fy + 1.f - iy);
            const float w10 = (fx + 1.f - ix) * (iy - fy), w11 = (ix - fx) * (iy - fy);
            const bool xa = (unsigned)x0 < (unsigned)p.W, xb = (unsigned)(x0 + 1) < (unsigned)p.W;
            const bool ya = (unsigned)y0 < (unsigned)p.H, yb = (unsigned)(y0 + 1) < (unsigned)p.H;
            const long o00 = (long)y0 * p.W + x0;
            const float gk = g[(size_t)k * cs];
#pragma unroll
            for (int c = 0; c < kBwdCh; ++c) {
                if (c >= nc) break;
                const float *pl = f1p + c * plane;
                float sv = 0.f;
                if (ya & xa) sv += pl[o00] * w00;
                if (ya & xb) sv += pl[o00 + 1] * w01;
                if (yb & xa) sv += pl[o00 + p.W] * w10;
                if (yb & xb) sv += pl[o00 + p.W + 1] * w11;
                acc[c] = fmaf(gk, sv, acc[c]);
            }
        }
        float *dst = gf0 + (size_t)b * gf0_bs + (size_t)c0 * cs + (size_t)i * p.G + j;
#pragma unroll
        for (int c = 0; c < kBwdCh; ++c)
            if (c < nc) dst[c * cs] = acc[c] / p.sqrt_c;
    }
}

#include "local_corr_stage.h"

// ---- fast tiled kernel -----------------------------------------------------------------------
// Stage traffic: wave-iteration wi covers channel group (wi & 3) of the 64 region pixels starting
// at (wi >> 2) * 64: four coalesced row-segment loads (one per channel) and one 16-byte LDS write
// per lane.  Issue and commit are separate so that a whole chunk's loads are in flight at once and
// the NEXT chunk's loads stay in flight across the D-stage.  Addresses are clamped instead of
// branched around (a conditional load becomes a branch + wait per element).
template <int N, typename FT>
struct StageRegs {
    FT v[N][4];  // as stored: fp16 values are widened at the commit (widening at the load makes hipcc wait for the loads in
                 // small groups instead of keeping a chunk's worth in flight)
    int dst[N];  // float4 index in the stage, or -1
};

template <int N, typename FT>
__device__ __forceinline__ void stage_issue(StageRegs<N, FT> &r, const FT *f1c, int H, int W, const Region &rg, int wave,
                                            int lane, int wi_begin) {
    const int npx = rg.w * rg.h;
    const int nwi = ((npx + 63) >> 6) * 4;
    const float inv_w = __builtin_amdgcn_rcpf((float)rg.w);  // 1 ulp is plenty: (q + 0.5) / w stays >= 0.5 / w away from an integer
    const unsigned pl32 = (unsigned)(H * W);
#pragma unroll
    for (int u = 0; u < N; ++u) {
        const int wi = wi_begin + wave + u * kWaves;
        const int cg = wi & 3;
        const int q = ((wi >> 2) << 6) + lane;
        const bool ok = (wi < nwi) & (q < npx);
        const int y = (int)(((float)q + 0.5f) * inv_w);  // exact for q < 2^16, w < 2^10
        const int x = q - y * rg.w;
        // 32-bit element offsets from the wave-uniform chunk base (C*H*W < 2^31 is checked on the host):
        // one VGPR per address instead of a 64-bit pair
        const unsigned off = ok ? (unsigned)(cg * 4) * pl32 + (unsigned)((rg.y0 + y) * W + (rg.x0 + x)) : 0u;
        const unsigned st = ok ? pl32 : 0u;
        r.v[u][0] = f1c[off];
        r.v[u][1] = f1c[off + st];
        r.v[u][2] = f1c[off + 2 * st];
        r.v[u][3] = f1c[off + 3 * st];
        r.dst[u] = ok ? (y * rg.pitch + x) * kSlotV4 + cg : -1;
    }
}

template <int N, typename FT>
__device__ __forceinline__ void stage_commit(float4 *s4, const StageRegs<N, FT> &r) {
#pragma unroll
    for (int u = 0; u < N; ++u)
        if (r.dst[u] >= 0) s4[r.dst[u]] = make_float4((float)r.v[u][0], (float)r.v[u][1], (float)r.v[u][2], (float)r.v[u][3]);
}

// whatever of the region the first `done` wave-iterations per wave did not cover
template <int N, typename FT>
__device__ __forceinline__ void stage_rest(float4 *s4, const FT *f1c, int H, int W, const Region &rg, int wave, int lane,
                                           int done) {
    const int nwi = ((rg.w * rg.h + 63) >> 6) * 4;
    for (int wi0 = done * kWaves; wi0 < nwi; wi0 += kWaves * N) {
        StageRegs<N, FT> r;
        stage_issue(r, f1c, H, W, rg, wave, lane, wi0);
        stage_commit(s4, r);
    }
}

// One tile of 2*ROUNDS x 16 cells.
//   STAGED = true : regular path -- the tile's windows are staged in LDS; a tile whose windows do
//                   not fit is appended to p.todo and left to the second launch.
//   STAGED = false: irregular path -- same arithmetic, patch pixels gathered straight from f1 (L2).
// Geometry: the tile's NC = 32*ROUNDS cells form a block TW cells wide whose top-left cell is
// (row0, col0) of image b's grid; only its first `rows` rows belong to it (sub-tiles of a 2-row tile).
//   SECOND = false: first launch (TW = 16, one tile per workgroup).
//   SECOND = true : second launch -- an irregular tile is cut into 4 x 8 sub-tiles, each staged on
//                   its own (half the footprint along the grid row, so twice the magnification
//                   fits); a sub-tile that still does not fit falls through to the gather variant.
//   QOK: the r >= 5 staging may use the 16-byte quad loads (fp16 maps: only with an even width -- 4-byte aligned 8-byte quads; the
//        host picks the instantiation.  As a run-time flag both staging forms' registers were live at once: the fp16 kernels spilled
//        16-39 registers at 128)
template <int R, int ROUNDS, bool STAGED, int TW, bool SECOND, typename FT, int STAGE = 68 * 1024, bool QOK = true>
__device__ __forceinline__ void process_tile(const LcParams &p, int b, int row0, int col0, int rows, unsigned wid,
                                             unsigned char *smem) {
    constexpr int kStageBytes = STAGE;       // shadow the file-level defaults: the lean kernel runs this path inside its own,
    constexpr int kCapSlots = STAGE / (kSlotV4 * 16) - 1;  // smaller LDS allocation
    constexpr int PW = 2 * R + 2;            // patch width: taps -R..R plus the +1 bilinear neighbour
    constexpr int P = PW * PW;               // patch positions per cell
    constexpr int NP = (P + 15) / 16;        // positions per lane
    constexpr int D = 2 * R + 1, K = D * D;
    constexpr int NC = 32 * ROUNDS;          // cells per tile (2 * ROUNDS rows of 16)
    constexpr int DS = P + 1;                // D-buffer cell stride (odd: conflict-free epilogue reads)
    constexpr int TS = 2 * D + 1;            // fraction-table cell stride (odd)
    // small windows (r <= 2, 16-channel maps) leave LDS for a fraction table of its own next to the stage: it is then filled
    // under the first chunk's stage loads instead of after the D-stage (9 % of the r = 2 kernel)
    constexpr bool kEarlyTab = R <= 2;
    static_assert((NC * DS + (kEarlyTab ? 0 : NC * TS)) * 4 <= kStageBytes, "D buffer (+ fraction table) must fit in the stage they alias");

    float4 *s4 = reinterpret_cast<float4 *>(smem);
    float *dbuf = reinterpret_cast<float *>(smem);
    int *cellX0 = reinterpret_cast<int *>(smem + kStageBytes);
    int *cellY0 = cellX0 + NC;
    float *cellNx = reinterpret_cast<float *>(cellY0 + NC);  // normalised centre (flow) of the cell
    float *cellNy = cellNx + NC;
    int *cellSlow = reinterpret_cast<int *>(cellNy + NC);    // [NC] 1 = redo this cell with the per-tap routine
    int *bbox = cellSlow + NC;                                // x0,y0,x1,y1 of the tile's windows
    int *nSlow = bbox + 4;                                    // number of flagged cells in the tile
    int *allInside = bbox + 5;                                // 1 = no window of the tile touches the image border
    constexpr int kCellBytes = (NC * 20 + 32 + 15) & ~15;
    constexpr int kTabBytes = kEarlyTab ? (NC * TS * 4 + 15) & ~15 : 0;
    float *tab = kEarlyTab ? reinterpret_cast<float *>(smem + kStageBytes + kCellBytes)  // [NC][TS] per-tap fractions
                           : dbuf + NC * DS;                                               // ... or aliasing the stage
    float *f0s = reinterpret_cast<float *>(smem + kStageBytes + kCellBytes + kTabBytes);  // [NC][C+4]: the tile's f0, cell-major
    const int CS = p.C + 4;                                   // +4: the 4 cells a wave reads hit different banks

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = p.G, H = p.H, W = p.W;
#ifdef GFN_ABLATE
    const bool stamping = ABL(p, 512) && blockIdx.x == 2000 && tid == 0 && !SECOND;
    long long stamp[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    STAMP(0);
    auto cell_gi = [&](int cell) { return row0 + cell / TW; };
    auto cell_gj = [&](int cell) { return col0 + cell % TW; };
    auto cell_ok = [&](int cell) { return (cell / TW < rows) & (row0 + cell / TW < G) & (col0 + cell % TW < G); };

    // ---- per-cell setup: patch origin, bounding box -------------------------------------------
    if (tid < 4) bbox[tid] = (tid & 2) ? -kFar : kFar;
    if (tid == 0) { *nSlow = 0; *allInside = 1; }
    __syncthreads();
    const float xhi = p.win_xhi, xlo = -xhi, yhi = p.win_yhi, ylo = -yhi;  // +-2r/W, +-2r/H as fp32
    // the cell's flow: requested before the f0 block below so that the two DRAM round trips of the set-up overlap
    float pre_nx = 0.f, pre_ny = 0.f;
    if (tid < NC && cell_ok(tid)) cell_coords(p, b, cell_gi(tid), cell_gj(tid), pre_nx, pre_ny);
    // the tile's f0 block (NC cells x C channels, 8-16 KB): coalesced 64-byte row segments -> LDS,
    // cell-major.  (Loading f0 per lane would issue 16 loads per round and chunk that fetch 16 bytes each.)
    {
        const float *f0b = p.f0 + (size_t)b * p.f0_bs;
        const int total = p.C * NC;
        constexpr int UF = 4;
        for (int e0 = tid; e0 < total; e0 += kThreads * UF) {
            float v[UF];
#pragma unroll
            for (int q = 0; q < UF; ++q) {
                const int e = e0 + q * kThreads;
                const int c = e / NC, cell = e - c * NC;
                const int gi = cell_gi(cell), gj = cell_gj(cell);
                const bool ok = (e < total) & cell_ok(cell) & !ABL(p, 4);
                v[q] = f0b[ok ? (size_t)c * G * G + (size_t)gi * G + gj : 0];  // clamped address, select below
                v[q] = ok ? v[q] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < UF; ++q) {
                const int e = e0 + q * kThreads;
                const int c = e / NC, cell = e - c * NC;
                if (e < total) f0s[cell * CS + c] = v[q];
            }
        }
    }
    STAMP(11);
    int bx0 = kFar, by0 = kFar, bx1 = -kFar, by1 = -kFar;  // this cell's window clipped to the image
    bool inside = true;                                      // ... and whether clipping changed nothing
    if (tid < NC) {
        int X0 = kFar, Y0 = kFar, slow = 0;
        float nx = 0.f, ny = 0.f;
        if (cell_ok(tid)) {
            nx = pre_nx; ny = pre_ny;
            // patch origin = floor of the reference's own fp32 coordinate of tap 0:
            // taps kx=0..2R then read columns kx and kx+1 of the patch
            const float fx = floorf(unnorm(nx + xlo, W));  // linspace(lo, hi, D)[0] == lo
            const float fy = floorf(unnorm(ny + ylo, H));
            if ((fx > -1e6f) & (fx < 1e6f) & (fy > -1e6f) & (fy < 1e6f)) {  // false for nan/inf
                X0 = (int)fx;
                Y0 = (int)fy;
                if (STAGED) {
                    const int x0 = max(X0, 0), x1 = min(X0 + PW, W), y0 = max(Y0, 0), y1 = min(Y0 + PW, H);
                    if (x0 < x1 && y0 < y1) { bx0 = x0; by0 = y0; bx1 = x1; by1 = y1; }
                    inside = (X0 >= 0) & (X0 + PW <= W) & (Y0 >= 0) & (Y0 + PW <= H);
                }
            } else {
                slow = 1;  // non-finite / absurd flow: let the per-tap routine decide
                atomicAdd(nSlow, 1);
                inside = false;
            }
        }
        cellX0[tid] = X0;
        cellY0[tid] = Y0;
        cellNx[tid] = nx;
        cellNy[tid] = ny;
        cellSlow[tid] = slow;
    }
    STAMP(12);
    if (STAGED && wave < (NC + 63) / 64) {  // bounding box: reduce inside the wave (all 64 lanes take
        // part, idle ones with the identity) with DPP row shifts / broadcasts -- the ds_bpermute butterfly took 1500+ cycles
        // of every tile's critical path -- then one LDS update per wave and bound
        bx0 = wave_min_i32(bx0); by0 = wave_min_i32(by0);
        bx1 = -wave_min_i32(-bx1); by1 = -wave_min_i32(-by1);
        // cells off the grid or with absurd flow have inside == true but an empty box: harmless
        const bool all_in = __all(inside || tid >= NC);
        if (lane == 0) {
            if (NC <= 64) {  // a single wave holds every cell: plain stores
                bbox[0] = bx0; bbox[1] = by0; bbox[2] = bx1; bbox[3] = by1;
            } else {
                atomicMin(bbox + 0, bx0);
                atomicMin(bbox + 1, by0);
                atomicMax(bbox + 2, bx1);
                atomicMax(bbox + 3, by1);
            }
            if (!all_in) *allInside = 0;
        }
    }
    // the zero slot (index kCapSlots) is what every out-of-image tap reads
    if (STAGED && tid < kSlotV4) s4[kCapSlots * kSlotV4 + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    STAMP(1);
    __syncthreads();
    STAMP(2);

    // ---- the staging region (block-uniform) -----------------------------------------------------
    Region u;
    u.x0 = bbox[0]; u.y0 = bbox[1];
    u.w = max(bbox[2] - u.x0, 0); u.h = max(bbox[3] - u.y0, 0);
    // Round 3, r >= 5 (64-channel maps): the stage loads are 16-byte quads through a buffer descriptor (local_corr_stage.h) instead
    // of 4-byte loads -- a 2 x 16-cell tile at r = 6 issued ~750 wave-level loads, the whole kernel's time in the CU's
    // vector-memory issue path.  A row is then whole quads (the last one may hang over the region's right edge: its pixels land in
    // pad slots nobody reads; past the tensor the descriptor returns zeros).  fp16 maps need even x0 and even rows (4-byte aligned
    // 8-byte quads); odd-width fp16 maps keep the narrow loads.
    constexpr bool kQuads = STAGED && R >= 5 && QOK;
    constexpr bool quads = kQuads;  // a compile-time constant: no branch around the loads, one staging form's registers
    if (quads && sizeof(FT) == 2 && (u.x0 & 1)) { u.x0 -= 1; u.w += 1; }
    const int w4 = quads ? ((u.w + 3) & ~3) : u.w;
    u.pitch = w4 + ((PW - w4) & 15);  // pitch == patch width (mod 16): conflict-free b128 reads across patch rows
    // a region that only fits without that padding is staged unpadded: some bank conflicts in the D-stage cost far less
    // than the second launch
    if (STAGED && (long)u.pitch * u.h > kCapSlots && (long)w4 * u.h <= kCapSlots) u.pitch = w4;
    if (STAGED && (long)u.pitch * u.h > kCapSlots) {
        // strong magnification / rotation / scattered flow: the windows do not fit the stage
        if (!SECOND) {
            if (tid == 0) p.todo[kTodoHdr + atomicAdd(p.todo, 1)] = (int)wid;
        } else {
            __syncthreads();
            if (ABL(p, 1024)) return;
            process_tile<R, ROUNDS, false, TW, true, FT, STAGE, QOK>(p, b, row0, col0, rows, wid, smem);  // gather from L2
        }
        return;
    }

    // the first chunk's stage loads go out before the per-lane addressing below: ~2.5 k cycles of index arithmetic under the
    // round trip instead of in front of it
    const FT *f1b = f1_of<FT>(p, b);
    constexpr int PRE = 4;  // wave-iterations of stage loads kept in flight (48 x 64 px x 4 ch = a 768-pixel region)
    StageRegs<PRE, FT> pre;
    constexpr int PRE0 = R <= 2 ? 4 : 6;  // the first chunk is requested before the D-stage registers exist: more of it in flight at
                                          // once (r <= 2 regions need 3.5 iterations; unused ones still cost their index math)
    StageRegs<PRE0, FT> pre0;
    // the quad form (r >= 5): the region as a RowPlan, two items per lane in flight
    RowPlan up;
    up.x0 = u.x0; up.y0 = u.y0; up.w = u.w; up.h = u.h; up.pitch = u.pitch; up.nq = (u.w + 3) >> 2;
    up.nitems = ((up.h * up.nq + 15) / 16 + 7) & ~7;
    QuadLane qlq;
    QuadRegs<kQuadPre, FT> preq;
    const rsrc_t f1r = make_rsrc(f1b, (unsigned)p.C * (unsigned)(H * W) * (unsigned)sizeof(FT));
    if (quads) {
#pragma unroll
        for (int n = 0; n < kQuadPre; ++n) qlq.it[n] = quad_item<false, FT, kSlotV4, true>(up, H, W, wave, lane, n);
        quad_issue<kQuadPre, false, FT, kSlotV4, true>(preq, f1r, 0u, H, W, up, wave, lane, qlq, 0);
    } else if (STAGED && !ABL(p, 1)) {
        stage_issue(pre0, f1b, H, W, u, wave, lane, 0);
    }

    // fraction table: the reference's fp32 coordinate of every tap column / row of every cell
    // (local_correlation.py:55 adds window offsets in normalised units, grid_sample un-normalises)
    auto fill_table = [&]() {
        for (int e = tid; e < NC * 2 * D && !ABL(p, 32); e += kThreads) {
            const int cell = e / (2 * D), a = e - cell * (2 * D);
            const bool isy = a >= D;
            const int k = isy ? a - D : a;
            const float n = isy ? cellNy[cell] : cellNx[cell];
            const float pix = unnorm(n + (isy ? gfn::linspace_step_at(ylo, yhi, p.win_ystep, D, k) : gfn::linspace_step_at(xlo, xhi, p.win_xstep, D, k)),
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
    };
    if (kEarlyTab) fill_table();

    // ---- per-lane D-stage addressing -----------------------------------------------------------
    int g, s16;
    lane_group(lane, g, s16);
    const int cr = wave * 4 + g;  // cell inside a round (0..31): row cr>>4, column cr&15
    const bool interior = STAGED && *allInside != 0;  // block-uniform
    unsigned apk[ROUNDS][(NP + 1) / 2];  // staged: float4 index (< 2^16) of each (round, pass) patch pixel, two per register
    float acc[ROUNDS][NP];
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
        const int cell = rd * 32 + cr;
        const int X0 = cellX0[cell], Y0 = cellY0[cell];
        // interior tiles (no window touches the border, every cell on the grid): no per-pixel tests
        const int base = (Y0 - u.y0) * u.pitch + (X0 - u.x0);
#pragma unroll
        for (int t = 0; t < NP; ++t) {
            acc[rd][t] = 0.f;
            if (STAGED) {
                const int pp = s16 + 16 * t;
                const int yy = pp / PW, xx = pp - yy * PW;
                int slot;
                if (interior) {
                    slot = (pp < P && X0 != kFar) ? base + yy * u.pitch + xx : kCapSlots;
                } else {
                    const int X = X0 + xx, Y = Y0 + yy;
                    const bool in = (pp < P) & ((unsigned)X < (unsigned)W) & ((unsigned)Y < (unsigned)H);
                    slot = in ? (Y - u.y0) * u.pitch + (X - u.x0) : kCapSlots;
                }
                const unsigned a = (unsigned)(slot * kSlotV4);
                if (t & 1)
                    apk[rd][t >> 1] |= a << 16;
                else
                    apk[rd][t >> 1] = a;
            }
        }
    }

    STAMP(3);
    // ---- main loop: 16 channels at a time ----------------------------------------------------
    const size_t cs = (size_t)G * G;
    if (quads) {
        quad_commit<kQuadPre, false, FT>(s4, preq, H, W, up, wave, lane, qlq, 0);
        quad_rest<false, FT>(s4, f1r, 0u, H, W, up, wave, lane, qlq, kQuadPre);
    } else if (STAGED && !ABL(p, 1)) {
        stage_commit(s4, pre0);
        stage_rest<2>(s4, f1b, H, W, u, wave, lane, PRE0);
    }
    if (STAGED) __syncthreads();
    STAMP(4);
    for (int c0 = 0; c0 < p.C; c0 += kChunk) {
        const FT *f1c = f1b + (size_t)c0 * H * W;
        const bool more = c0 + kChunk < p.C;
        if (STAGED) {
            // keep the packed indices packed: without this the unpacking is hoisted out of the loop and
            // the unpacked copies cost NP more registers per round
#pragma unroll
            for (int rd = 0; rd < ROUNDS; ++rd)
#pragma unroll
                for (int h = 0; h < (NP + 1) / 2; ++h) asm volatile("" : "+v"(apk[rd][h]));
            // next chunk's loads: in flight across this chunk's D-stage
            const unsigned next_off = (unsigned)(c0 + kChunk) * (unsigned)(H * W) * (unsigned)sizeof(FT);
            if (more && quads) quad_issue<kQuadPre, false, FT, kSlotV4, true>(preq, f1r, next_off, H, W, up, wave, lane, qlq, 0);
            else if (more && !ABL(p, 1)) stage_issue(pre, f1c + (size_t)kChunk * H * W, H, W, u, wave, lane, 0);
        }
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            float f[kChunk];
            {   // this lane's cell, 16 channels: 4 LDS reads (the 16 lanes of a cell read the same address)
                const float4 *fq = reinterpret_cast<const float4 *>(f0s + (rd * 32 + cr) * CS + c0);
                const float4 a0 = fq[0], a1 = fq[1], a2 = fq[2], a3 = fq[3];
                f[0] = a0.x; f[1] = a0.y; f[2] = a0.z; f[3] = a0.w; f[4] = a1.x; f[5] = a1.y; f[6] = a1.z; f[7] = a1.w;
                f[8] = a2.x; f[9] = a2.y; f[10] = a2.z; f[11] = a2.w; f[12] = a3.x; f[13] = a3.y; f[14] = a3.z; f[15] = a3.w;
            }
            if (ABL(p, 2)) continue;
            if (STAGED) {
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
            } else {
                const size_t plane = (size_t)H * W;
                const int cX0 = cellX0[rd * 32 + cr], cY0 = cellY0[rd * 32 + cr];
                // TG patch positions x 16 channels = 64 gathers in flight per lane: this path is pure L2 latency
                // (one position at a time left a sub-tile of scattered flow at 150-200 us)
                constexpr int TG = 4;
#pragma unroll
                for (int t0 = 0; t0 < NP; t0 += TG) {
                    FT v[TG][kChunk];  // as stored: fp16 values are widened at their use (widening at the load made hipcc wait for
                                       // the gathers in small groups: the scattered-flow path ran 2x slower on fp16 maps)
                    bool in[TG];
#pragma unroll
                    for (int tt = 0; tt < TG; ++tt) {
                        if (t0 + tt < NP) {
                            const int pp = s16 + 16 * (t0 + tt);
                            const int yy = pp / PW, xx = pp - yy * PW;
                            const int X = cX0 + xx, Y = cY0 + yy;
                            in[tt] = (pp < P) & ((unsigned)X < (unsigned)W) & ((unsigned)Y < (unsigned)H) & !ABL(p, 2048);
                            const unsigned off = in[tt] ? (unsigned)(Y * W + X) : 0u;  // offset 0 when outside: valid memory, masked below
#pragma unroll
                            for (int k = 0; k < kChunk; ++k) v[tt][k] = f1c[k * plane + off];  // scalar plane base + 32-bit lane offset
                        }
                    }
#pragma unroll
                    for (int tt = 0; tt < TG; ++tt) {
                        if (t0 + tt < NP) {
                            float a = acc[rd][t0 + tt];
#pragma unroll
                            for (int k = 0; k < kChunk; ++k) a = fmaf(f[k], in[tt] ? (float)v[tt][k] : 0.f, a);
                            acc[rd][t0 + tt] = a;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);  // one group of loads in flight at a time (register budget)
                }
            }
        }
        // do not let the scheduler sink this chunk's FMAs below the barrier (it would keep every
        // LDS read result of the chunk live across it: hundreds of spills)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd)
#pragma unroll
            for (int t = 0; t < NP; ++t) asm volatile("" : "+v"(acc[rd][t]));  // pins the FMAs above this point
        STAMP(5 + (c0 != 0 ? 2 : 0));
        if (STAGED && more) {
            __syncthreads();  // everyone is done reading this chunk
            if (quads) {
                const unsigned next_off = (unsigned)(c0 + kChunk) * (unsigned)(H * W) * (unsigned)sizeof(FT);
                quad_commit<kQuadPre, false, FT>(s4, preq, H, W, up, wave, lane, qlq, 0);
                quad_rest<false, FT>(s4, f1r, next_off, H, W, up, wave, lane, qlq, kQuadPre);
            } else if (!ABL(p, 1)) {
                stage_commit(s4, pre);
                stage_rest<2>(s4, f1c + (size_t)kChunk * H * W, H, W, u, wave, lane, PRE);
            }
            __syncthreads();
            STAMP(6);
        }
    }

    // ---- epilogue: D -> LDS, per-tap fractions, bilinear combination, coalesced stores -------
    __syncthreads();
    STAMP(8);
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
        const int cell = rd * 32 + cr;
#pragma unroll
        for (int t = 0; t < NP; ++t) {
            const int pp = s16 + 16 * t;
            if (pp < P) dbuf[cell * DS + pp] = acc[rd][t];
        }
    }
    if (!kEarlyTab) fill_table();
    __syncthreads();
    STAMP(9);
    {
        // each wave combines CW cells x a strided subset of the K taps: lane -> cell (so that stores run
        // along the grid row), taps strided over the waves (and over lane halves when NC == 32)
        constexpr int CW = NC >= 64 ? 64 : 32;          // cells per wave-instruction
        constexpr int NCB = NC / CW;                    // groups of CW cells per tile
        constexpr int WPB = kWaves / NCB * (64 / CW);   // tap phases sharing one group of cells
        const int cell = (wave / (kWaves / NCB)) * CW + (lane & (CW - 1));
        const int kphase = (wave % (kWaves / NCB)) * (64 / CW) + (lane / CW);
        const int gi = cell_gi(cell), gj = cell_gj(cell);
        if (cell_ok(cell) && !cellSlow[cell] && !ABL(p, 8)) {
            // one tap ROW (ky) at a time: the two D rows it needs are read once (2*PW LDS reads for D
            // outputs), the column fractions of the cell stay in registers
            const float *dc = dbuf + cell * DS;
            const float *tc = tab + cell * TS;
            float *o = p.out + (size_t)b * p.out_bs + (size_t)gi * G + gj;
            float wx1[D], wx0[D];
#pragma unroll
            for (int kx = 0; kx < D; ++kx) { wx1[kx] = tc[kx]; wx0[kx] = 1.f - wx1[kx]; }
            constexpr int NR = (D + WPB - 1) / WPB;
#pragma unroll
            for (int n = 0; n < NR; ++n) {
                const int ky = kphase + n * WPB;
                if (ky < D) {
                    // separable bilinear: the PW patch columns are blended vertically once (1/sqrt(C) folded into the row
                    // weights), every tap is then two instructions -- 4 per output instead of the 12 of the four-corner
                    // form (the combine was ~28 % of a tile's vector instructions at r = 4)
                    const float wy1 = tc[D + ky];
                    const float wy1s = wy1 * p.inv_sqrt_c, wy0s = (1.f - wy1) * p.inv_sqrt_c;
                    const float *d = dc + ky * PW;
                    float m[PW];
#pragma unroll
                    for (int x = 0; x < PW; ++x) m[x] = fmaf(d[PW + x], wy1s, d[x] * wy0s);
                    float *ok = o + (size_t)(ky * D) * cs;
#pragma unroll
                    for (int kx = 0; kx < D; ++kx)
                        __builtin_nontemporal_store(fmaf(m[kx + 1], wx1[kx], m[kx] * wx0[kx]),
                                                    ok + (size_t)kx * cs);  // streamed: nothing on the hot path reads it back
                }
            }
        }
    }

    STAMP(10);
#ifdef GFN_ABLATE
    if (stamping)
        printf("stamps(cycles from entry): flow-issued+f0 staged %lld | cells done %lld | setup-done %lld | barrier %lld | addressing %lld | chunk0 staged %lld | D chunk0 %lld | chunk1 staged %lld | "
               "D chunk1 %lld | barrier %lld | dbuf+table %lld | combine+stores issued %lld\n",
               stamp[11] - stamp[0], stamp[12] - stamp[0], stamp[1] - stamp[0], stamp[2] - stamp[0], stamp[3] - stamp[0], stamp[4] - stamp[0], stamp[5] - stamp[0], stamp[6] - stamp[0],
               stamp[7] - stamp[0], stamp[8] - stamp[0], stamp[9] - stamp[0], stamp[10] - stamp[0]);
#endif
    // ---- flagged cells: general per-tap routine (about one cell in 10^4) ------------------------
    if (*nSlow != 0 && !ABL(p, 16)) {  // block-uniform, rare
        // compact the flagged cells (cellX0 is free now), then spread (cell, tap) pairs over the whole workgroup
        __syncthreads();
        if (tid == 0) {
            int n = 0;
            for (int cell = 0; cell < NC; ++cell)
                if (cellSlow[cell] && cell_ok(cell)) cellX0[n++] = cell;
            *nSlow = n;
        }
        __syncthreads();
        const int total = *nSlow * K;
        for (int e = tid; e < total; e += kThreads) {
            const int cell = cellX0[e / K], k = e % K;
            const int gi = cell_gi(cell), gj = cell_gj(cell);
            p.out[(size_t)b * p.out_bs + ((size_t)k * G + gi) * G + gj] =
                tap_general<FT>(p, b, gi, gj, k / D, k % D, D, cellNx[cell], cellNy[cell]);
        }
    }
}

template <int R, int ROUNDS, typename FT, bool QOK = true>
__global__ __launch_bounds__(kThreads, 4) void local_corr_tile_kernel(LcParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef GFN_ABLATE
    {   // experiment: de-synchronise the two workgroups of a CU (first dispatch wave only)
        const unsigned bid = blockIdx.x;
        const bool late = (ABL(p, 64) && ((bid >> 8) & 1)) || (ABL(p, 128) && (bid & 1)) || (ABL(p, 256) && ((bid >> 3) & 1));
        if (late && bid < 512) {
            const int n = p.dbg >> 12;  // sleep units of ~64*127 cycles
            for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
        }
    }
#endif
    const unsigned wid = gfn::xcd_remap(blockIdx.x, gridDim.x);
    const int tiles = p.tiles_x * p.tiles_y;
    const int b = wid / tiles, tile = wid - b * tiles;
    const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    process_tile<R, ROUNDS, true, kTileW, false, FT, 68 * 1024, QOK>(p, b, ty * 2 * ROUNDS, tx * kTileW, 2 * ROUNDS, wid, smem);
}

// second launch: the tiles the staged kernel left in p.todo (their number is only known on the
// device), re-cut into sub-tiles 8 cells wide and up to 4 rows high
// worker `me` of `nworkers`: the separate second launch (round-1 path: a workgroup of its own), or the first workgroups of the
// lean tile kernel (which finishes the plan launch's list inside its own launch, with its own -- smaller -- stage)
template <int R, int ROUNDS, typename FT, int STAGE, bool QOK = true>
__device__ __forceinline__ void second_launch_worker(const LcParams &p, unsigned char *smem, int me, int nworkers) {
    constexpr int TH = 2 * ROUNDS, SH = TH < 4 ? TH : 4;  // sub-tile height
    constexpr int SUBS = (TH / SH) * (kTileW / 8);
    const int n = p.todo[0] * SUBS;
    const int tiles = p.tiles_x * p.tiles_y;
    // Only as many workgroups as there are work items take part (the rest leave at once): with no tile on the list -- the
    // common case -- the launch costs one load per workgroup instead of two contended atomics (14 us for 512 workgroups).
    const int part = n < nworkers ? n : nworkers;
    if (me >= (part > 0 ? part : 1)) return;
    // work items differ by 10x (a staged sub-tile vs one that gathers from L2): after its first item (its own block id) a
    // workgroup draws tickets first come, first served
    __shared__ int next_item;
    int it = me;
    while (it < n) {
        const unsigned wid = (unsigned)p.todo[kTodoHdr + it / SUBS];
        const int sub = it % SUBS;
        const int b = wid / tiles, tile = wid - b * tiles;
        const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
        process_tile<R, 1, true, 8, true, FT, STAGE, QOK>(p, b, ty * TH + (sub >> 1) * SH, tx * kTileW + (sub & 1) * 8, SH, wid, smem);
        __syncthreads();  // LDS (and next_item) are reused by the next sub-tile
        if (threadIdx.x == 0) next_item = part + atomicAdd(p.todo + 1, 1);
        __syncthreads();
        it = __builtin_amdgcn_readfirstlane(next_item);  // scalar: everything derived from it (b, map bases) stays in SGPRs
    }
    // the last workgroup to leave puts the counters back to zero for the next call: no memset node in front of every call.
    // Every participant has read todo[0] and drawn its last queue ticket before it gets here.
    if (threadIdx.x == 0 && (part <= 1 || atomicAdd(p.todo + 2, 1) == part - 1)) {
        p.todo[3] = p.todo[0];  // informational (tools/count_irregular.py, bench.py)
        p.todo[5] = p.todo[4];
        p.todo[4] = 0;
        p.todo[7] = p.todo[6];
        p.todo[6] = 0;
        p.todo[0] = 0;
        p.todo[1] = 0;
        p.todo[2] = 0;
    }
}

template <int R, int ROUNDS, typename FT, bool QOK = true>
__global__ __launch_bounds__(kThreads, 2) void local_corr_irregular_kernel(LcParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    second_launch_worker<R, ROUNDS, FT, kStageBytes, QOK>(p, smem, (int)blockIdx.x, (int)gridDim.x);
}

#include "local_corr_lean.h"
#include "local_corr_mstage.h"
#include "local_corr_mq.h"

// shapes the lean tile path takes (it keeps at most 8 channels of the f0 block per wave in registers, addresses planes with
// 32-bit byte offsets, and reads fp16 quads at 4-byte alignment)
bool lean_shape(int C, int H, int W, int G, int r, int f16) {
    const long K = (long)(2 * r + 1) * (2 * r + 1);
    if (!(r >= 1 && r <= 7 && (C == 16 || C == 32 || C == 64) && !(f16 && (W & 1)) && (long)C * H * W < (1L << 30) && K * G * G < (1L << 30) &&
          (long)C * G * G < (1L << 30)))
        return false;
    // Large windows (r >= 5): the lean tile kernel handles them (f0 block staged chunk by chunk, 65 792-byte stage at r = 7) and is on
    // par with the round-1 kernel on smooth flows where the 4 x 16-cell tile's region fits the stage (r = 7 on 32 x 32 maps: 80.7 vs
    // 79.5 us), slower where every tile would be staged in halves (r = 6 at spacing 1.75: 113 vs 79 us) and under the raw
    // soft-argmax flows scale 16 sees in the bench (160 vs 110 + 44 us: its in-launch second-launch workers) -- the round-1 kernel
    // with quad staging keeps them.  -DGFN_LEAN_R7=1 builds send r = 7 here.
    if (r >= 5) {
#if defined(GFN_LEAN_R7) && GFN_LEAN_R7
        const double sx = (double)W / G, sy = (double)H / G, pw = 2 * r + 2;
        return (3 * sy + pw + 1) * (15 * sx + pw + 4) <= 0.95 * 819;
#else
        return false;
#endif
    }
    return true;
}

// large windows on 64-channel maps: the matrix-core tile kernel of local_corr_mq.h is the first launch (byte offsets into the maps and
// the kOffRange marker share 31 bits)
bool mq_shape(int C, int H, int W, int G, int r, int f16) {
    const long K = (long)(2 * r + 1) * (2 * r + 1);
    return GFN_MQ != 0 && r >= 5 && r <= 7 && C == 64 && !(f16 && (W & 1)) && (long)C * H * W * (f16 ? 2 : 4) < 0x7FFFFFF0L &&
           K * G * G * 4 < 0x7FFFFFF0L && (long)C * G * G * 4 < 0x7FFFFFF0L;
}

// scratch layout of the lean path: header and tile list (ints), then the plan, 32-byte aligned
int *lean_plan_ptr(void *scratch, int B, int G) {
    const int64_t tiles_max = (int64_t)((G + 1) / 2) * ((G + 15) / 16) * B;  // what gfn_local_corr_scratch_bytes sized the list for
    return reinterpret_cast<int *>(((uintptr_t)scratch + 4 * (tiles_max + kTodoHdr) + 31) & ~(uintptr_t)31);
}

// the window parameters every tiled launch needs (what tap_general and the plan read)
template <int R>
void lean_window_params(LcParams &p) {
    p.tiles_x = (p.G + kTileW - 1) / kTileW;
    p.tiles_y = (p.G + 3) / 4;
    p.r = R; p.win_h = p.H; p.win_w = p.W; p.grid_based = 0;
    p.win_xhi = (float)(2.0 * R / p.W);
    p.win_yhi = (float)(2.0 * R / p.H);
    const volatile float xlo = -p.win_xhi, ylo = -p.win_yhi, n1 = (float)(2 * R);
    p.win_xstep = (p.win_xhi - xlo) / n1;
    p.win_ystep = (p.win_yhi - ylo) / n1;
}

// compute units of the current device (queried once per device: hipGetDeviceProperties is slow)
int device_cu_count() {
    static int cache[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cache[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cache[dev] = n;
    }
    return cache[dev];
}

template <int R, int NCH, typename FT>
void launch_lean(const LcParams &p, unsigned total, size_t lds, hipStream_t stream) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(local_corr_tile2_kernel<R, NCH, FT>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
    hipLaunchKernelGGL((local_corr_tile2_kernel<R, NCH, FT>), dim3(total + lean_workers<R>()), dim3(kThreads), lds, stream, p);
}

// the round-1 tile path: the staged tile kernel (r >= 5 on 64-channel maps: the matrix-core kernel of local_corr_mq.h) and the second
// launch for the tiles it listed
template <int R, int ROUNDS, typename FT, bool QOK>
int launch_tiles_staged(const LcParams &p, unsigned total, size_t lds, hipStream_t stream) {
    // per call and unconditional: the attribute is per device, a process may drive several
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(local_corr_tile_kernel<R, ROUNDS, FT, QOK>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(local_corr_irregular_kernel<R, ROUNDS, FT, QOK>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
    bool mq = false;
    if constexpr (R >= 5 && GFN_MQ != 0) {
        static_assert(ROUNDS == 1, "the matrix-core kernel's tiles are the round-1 kernel's 2 x 16 cells: they share the second launch");
        mq = p.mq != 0;
        if (mq) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(local_corr_mq_kernel<R, 64, FT, QOK>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kMqLds);
            hipLaunchKernelGGL((local_corr_mq_kernel<R, 64, FT, QOK>), dim3(total), dim3(kThreads), kMqLds, stream, p);
            if (int e = gfn::check_launch("local_corr_mq_kernel")) return e;
        }
    }
    if (!mq) {
        hipLaunchKernelGGL((local_corr_tile_kernel<R, ROUNDS, FT, QOK>), dim3(total), dim3(kThreads), lds, stream, p);
        if (int e = gfn::check_launch("local_corr_tile_kernel")) return e;
    }
    const unsigned grid2 = total < 256 ? total : 256;  // one per CU: with nothing on the list (the common case) the launch is pure overhead
    hipLaunchKernelGGL((local_corr_irregular_kernel<R, ROUNDS, FT, QOK>), dim3(grid2), dim3(kThreads), lds, stream, p);
    return gfn::check_launch("local_corr_irregular_kernel");
}

// the lean path (local_corr_lean.h): plan (unless the refiner-input launch wrote it), the tile kernel, and for r <= 2 the separate
// second launch (r >= 3: the tile kernel's first workgroups are the second launch)
template <int R, typename FT>
int launch_lean_path(const LcParams &p0, hipStream_t stream) {
    LcParams p = p0;
    lean_window_params<R>(p);
    constexpr int NC = 64;
    const unsigned total = (unsigned)p.B * p.tiles_x * p.tiles_y;
    if ((size_t)p.todo_ints < (size_t)total + kTodoHdr) return gfn::fail(GFN_ERR_SCRATCH, "local_corr: scratch too small");
    const size_t lds2 = Lean<R>::kStageLds + ((NC * 20 + 32 + 15) & ~15) + (Lean<R>::kTabInStage ? 0 : Lean<R>::kTabBytes) +  // (the table: inside the stage at r = 3, 4)
                        (size_t)NC * ((Lean<R>::kF0Chunk ? kChunk : p.C) + 4) * 4;
    if (lds2 > kMaxLds) return -1000;
    if (!p.planned) {
        hipLaunchKernelGGL((local_corr_plan_kernel<R>), dim3((total + 4 * kPlanPerWave - 1) / (4 * kPlanPerWave)), dim3(256), 0, stream, p);
        if (int e = gfn::check_launch("local_corr_plan_kernel")) return e;
    }
    auto tiles = [&]() {
        switch (p.C) {  // the lean kernel is specialised on the number of 16-channel chunks
            case 16: launch_lean<R, 1, FT>(p, total, lds2, stream); break;
            case 32: launch_lean<R, 2, FT>(p, total, lds2, stream); break;
            default: launch_lean<R, 4, FT>(p, total, lds2, stream); break;
        }
        return gfn::check_launch("local_corr_tile2_kernel");
    };
    auto second = [&]() {
        const size_t lds = kStageBytes + ((NC * 20 + 32 + 15) & ~15) + (R <= 2 ? ((NC * (2 * (2 * R + 1) + 1) * 4 + 15) & ~15) : 0) +
                           (size_t)NC * (p.C + 4) * 4;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(local_corr_irregular_kernel<R, 2, FT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  kMaxLds);
        const unsigned grid2 = total < 256 ? total : 256;  // one per CU: with nothing on the list (the common case) the launch is pure overhead
        hipLaunchKernelGGL((local_corr_irregular_kernel<R, 2, FT>), dim3(grid2), dim3(kThreads), lds, stream, p);
        return gfn::check_launch("local_corr_irregular_kernel");
    };
    if (lean_workers<R>() > 0) return tiles();  // its first workgroups are the second launch
    // (Round 5: marking the tile kernel hipExtAnyOrderLaunch behind a second launch issued first -- the two are independent, the plan
    // launch wrote the list -- does not make them run side by side on gfx950: 107.2 vs 108.5 us with three tiles listed, the flag is
    // documented as unsupported on GFX9.)
    if (int e = tiles()) return e;
    return second();
}

template <int R, int ROUNDS, typename FT>
int launch_tile(const LcParams &p0, hipStream_t stream, bool lean) {
#if defined(GFN_LEAN_R7) && GFN_LEAN_R7
    constexpr bool kLeanBuilt = true;
#else
    constexpr bool kLeanBuilt = R <= 4;  // lean_shape() never takes r >= 5 in this build: do not instantiate those kernels
#endif
    if constexpr (kLeanBuilt) {
        if (lean) {
            const int rc = launch_lean_path<R, FT>(p0, stream);
            if (rc != -1000) return rc;  // -1000: the lean kernel's LDS does not take this shape -> round-1 kernel below
        }
    }
    LcParams p = p0;
    constexpr int NC = 32 * ROUNDS;
    p.tiles_x = (p.G + kTileW - 1) / kTileW;
    p.tiles_y = (p.G + 2 * ROUNDS - 1) / (2 * ROUNDS);
    p.r = R; p.win_h = p.H; p.win_w = p.W; p.grid_based = 0;  // what tap_general needs for flagged cells
    p.win_xhi = (float)(2.0 * R / p.W);
    p.win_yhi = (float)(2.0 * R / p.H);
    {   // exactly the fp32 operations linspace_at performs: (end - start) / (float)(steps - 1)
        const volatile float xlo = -p.win_xhi, ylo = -p.win_yhi, n1 = (float)(2 * R);
        p.win_xstep = (p.win_xhi - xlo) / n1;
        p.win_ystep = (p.win_yhi - ylo) / n1;
    }
    const size_t lds = kStageBytes + ((NC * 20 + 32 + 15) & ~15) + (R <= 2 ? ((NC * (2 * (2 * R + 1) + 1) * 4 + 15) & ~15) : 0) +
                       (size_t)NC * (p.C + 4) * 4;
    // <= 80 KB (two workgroups per CU) for every shape GFNet uses; other C/r combinations still run,
    // one workgroup per CU; absurdly wide features go to the general kernel
    if (lds > kMaxLds) return -1000;
    // r >= 5 stages through a 32-bit buffer descriptor (local_corr_stage.h): maps of 2 GiB and more would wrap its byte count and
    // stage zeros; the general kernel (64-bit pointers) takes them
    if (R >= 5 && (long)p.C * p.H * p.W * (long)sizeof(FT) >= 0x7FFFFFF0L) return -1000;
    const unsigned total = (unsigned)p.B * p.tiles_x * p.tiles_y;
    if ((size_t)p.todo_ints < (size_t)total + kTodoHdr) return gfn::fail(GFN_ERR_SCRATCH, "local_corr: scratch too small");
    // fp16 maps of odd width (35 x 35: scale 8 of a 280-pixel refinement pass) keep the narrow stage loads at r >= 5: an 8-byte quad
    // of such a row is not 4-byte aligned.  A kernel instantiation of its own, so that the staging form is a compile-time constant.
    if constexpr (R >= 5 && sizeof(FT) == 2) {
        if (p.W & 1) return launch_tiles_staged<R, ROUNDS, FT, false>(p, total, lds, stream);
    }
    return launch_tiles_staged<R, ROUNDS, FT, true>(p, total, lds, stream);
}

}  // namespace

// ints of scratch the tiled path wants: the list of tiles left to the irregular launch.
GFN_EXPORT int64_t gfn_local_corr_scratch_bytes(int B, int G) {
    // smallest tile is 2 x 16 cells -> at most B * ceil(G/2) * ceil(G/16) tiles, plus the counter
    const int64_t tiles = (int64_t)((G + 1) / 2) * ((G + 15) / 16);
    // header + tile list (ints), then the lean path's plan (16 bytes per tile, 16-byte aligned)
    return 4 * ((int64_t)B * tiles + kTodoHdr) + 32 + 4 * kPlanInts * (int64_t)B * tiles;
}

GFN_EXPORT int gfn_local_corr_fwd_ex(const float *f0, int64_t f0_bs, const float *f1, const float *f1_second,
                                     const float *flow, float *out, int64_t out_bs, int B, int C, int G, int H, int W,
                                     int r, int grid_based, int win_h, int win_w, int variant, void *scratch,
                                     int64_t scratch_bytes, gfn_stream_t stream) {
    return gfn_local_corr_fwd_dt(f0, f0_bs, f1, f1_second, GFN_F32, flow, out, out_bs, B, C, G, H, W, r, grid_based, win_h, win_w, variant, scratch,
                                 scratch_bytes, stream);
}

GFN_EXPORT int gfn_local_corr_fwd_dt(const float *f0, int64_t f0_bs, const void *f1, const void *f1_second, int f1_dtype,
                                     const float *flow, float *out, int64_t out_bs, int B, int C, int G, int H, int W,
                                     int r, int grid_based, int win_h, int win_w, int variant, void *scratch,
                                     int64_t scratch_bytes, gfn_stream_t stream) {
    if (f1_dtype != GFN_F32 && f1_dtype != GFN_F16) return gfn::fail(GFN_ERR_INVALID_ARG, "local_corr: feature dtype must be GFN_F32 or GFN_F16");
    if (!f0 || !f1 || !out) return gfn::fail(GFN_ERR_INVALID_ARG, "local_corr: null tensor pointer");
    if (f1_second && (B & 1)) return gfn::fail(GFN_ERR_INVALID_ARG, "local_corr: symmetric batch must be even");
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
    p.f1_second = f1_second;
    p.f16 = f1_dtype == GFN_F16;
    p.Bh = f1_second ? B / 2 : B;
    p.f0_bs = f0_bs; p.out_bs = out_bs;
    p.B = B; p.C = C; p.G = G; p.H = H; p.W = W;
    p.tiles_x = p.tiles_y = 0;
    p.sqrt_c = (float)sqrt((double)C);
    p.inv_sqrt_c = (float)(1.0 / sqrt((double)C));
    p.r = r; p.win_h = win_h; p.win_w = win_w; p.grid_based = grid_based;
    p.todo = reinterpret_cast<int *>(scratch);
    p.todo_ints = scratch ? scratch_bytes / 4 : 0;
    p.plan = nullptr;
#ifdef GFN_ABLATE
    p.dbg = variant >> 8;
    variant &= 0xff;
#endif

    // the tiled path needs the tile list in scratch; without it the general kernel still gives the right answer
    // variant 0: the tiled path (lean fp32 tile kernel for r <= 4, the matrix-core kernel for r >= 5 on 64-channel maps); 1: general
    // kernel; 2: the round-1 tile kernel for every radius; 4: the lean tile kernel with the round-2 fp32 FMA D-stage (bit-identical
    // to variant 2; the fp32 cross-check / opt-out of the r >= 5 matrix-core kernel)
    const bool planned = (variant & 8) != 0;  // gfn_refiner_input_plan_fwd_dt has already written this call's plan
    variant &= ~8;
    const bool want_mm = variant == 0;
    if (variant == 4) variant = 0;
    const bool tiled = variant == 0 && flow && !grid_based && win_h == H && win_w == W;
    p.mq = (want_mm && tiled && mq_shape(C, H, W, G, r, p.f16)) ? 1 : 0;
    bool lean = tiled && lean_shape(C, H, W, G, r, p.f16);
    if (planned && !lean) return gfn::fail(GFN_ERR_INVALID_ARG, "local_corr: variant 8 (plan present) on a call the lean path does not take");
    p.planned = planned ? 1 : 0;
    if (lean && scratch) {
        p.plan = lean_plan_ptr(scratch, B, G);
        p.todo_ints = (int64_t)((G + 1) / 2) * ((G + 15) / 16) * B + kTodoHdr;
    }
    const bool fast_ok = (variant == 0 || variant == 2) && !grid_based && win_h == H && win_w == W && (C % kChunk) == 0 && r >= 1 && r <= 7 &&
                         scratch && scratch_bytes >= gfn_local_corr_scratch_bytes(B, G) && ((uintptr_t)scratch & 3) == 0;
    if (fast_ok) {
        int rc = -1000;
        if (p.f16) {
            switch (r) {
                case 1: rc = launch_tile<1, 2, _Float16>(p, s, lean); break;
                case 2: rc = launch_tile<2, 2, _Float16>(p, s, lean); break;
                case 3: rc = launch_tile<3, 2, _Float16>(p, s, lean); break;
                case 4: rc = launch_tile<4, 2, _Float16>(p, s, lean); break;
                case 5: rc = launch_tile<5, 1, _Float16>(p, s, lean); break;
                case 6: rc = launch_tile<6, 1, _Float16>(p, s, lean); break;
                case 7: rc = launch_tile<7, 1, _Float16>(p, s, lean); break;
            }
        } else {
            switch (r) {
                case 1: rc = launch_tile<1, 2, float>(p, s, lean); break;
                case 2: rc = launch_tile<2, 2, float>(p, s, lean); break;
                case 3: rc = launch_tile<3, 2, float>(p, s, lean); break;
                case 4: rc = launch_tile<4, 2, float>(p, s, lean); break;
                case 5: rc = launch_tile<5, 1, float>(p, s, lean); break;
                case 6: rc = launch_tile<6, 1, float>(p, s, lean); break;
                case 7: rc = launch_tile<7, 1, float>(p, s, lean); break;
            }
        }
        if (rc != -1000) return rc;  // -1000: shape not supported by the tiled path
    }
    const long total = (long)B * K * G * G;
    const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (p.f16) hipLaunchKernelGGL(local_corr_general_kernel<_Float16>, dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(local_corr_general_kernel<float>, dim3(grid), dim3(256), 0, s, p);
    return gfn::check_launch("local_corr_general_kernel");
}

GFN_EXPORT int gfn_local_corr_fwd(const float *f0, int64_t f0_bs, const float *f1, const float *f1_second,
                                  const float *flow, float *out, int64_t out_bs, int B, int C, int G, int H, int W, int r,
                                  int grid_based, int win_h, int win_w, void *scratch, int64_t scratch_bytes,
                                  gfn_stream_t stream) {
    return gfn_local_corr_fwd_ex(f0, f0_bs, f1, f1_second, flow, out, out_bs, B, C, G, H, W, r, grid_based, win_h, win_w, 0,
                                 scratch, scratch_bytes, stream);
}

// ---- refiner input + plan in one launch ----------------------------------------------------------------------------------
GFN_EXPORT int gfn_local_corr_plans(int C, int H, int W, int G, int r, int f1_dtype) {
    return lean_shape(C, H, W, G, r, f1_dtype == GFN_F16) ? 1 : 0;
}

namespace {
template <int R, typename FT>
int launch_ri_plan(const gfn_ri::RiArgs &q, LcParams p, hipStream_t s, bool keep) {
    lean_window_params<R>(p);
    const unsigned q_blocks = (unsigned)(((long)q.G * q.G + 255) / 256);
    const unsigned tiles = (unsigned)(p.tiles_x * p.tiles_y);
    const unsigned p_blocks = (tiles + 4 * kPlanPerWave - 1) / (4 * kPlanPerWave);
    const int banded = gfn_ri::ri_bands(q.B, q.Bh, q_blocks) ? 1 : 0;  // symmetric batches: XCD-banded order of the refiner-input blocks
    const dim3 grid = banded ? dim3(gfn_ri::ri_banded_blocks(q.B, q_blocks) + p_blocks * (unsigned)q.B) : dim3(q_blocks + p_blocks, (unsigned)q.B);
    if (keep) hipLaunchKernelGGL((refiner_input_plan_kernel<R, FT, true>), grid, dim3(256), 0, s, q, p, q_blocks, p_blocks, banded);
    else hipLaunchKernelGGL((refiner_input_plan_kernel<R, FT, false>), grid, dim3(256), 0, s, q, p, q_blocks, p_blocks, banded);
    return gfn::check_launch("refiner_input_plan_kernel");
}
}  // namespace

GFN_EXPORT int gfn_refiner_input_plan_fwd_dt(const void *f0, const void *f1, int dtype, const float *flow, const float *disp_w,
                                             const float *disp_b, float *d, int64_t d_bs, int B, int C, int Hs, int Ws, int G, int disp_dim,
                                             float disp_scale, int symmetric, int r, void *scratch, int64_t scratch_bytes,
                                             gfn_stream_t stream) {
    if (dtype != GFN_F32 && dtype != GFN_F16) return gfn::fail(GFN_ERR_INVALID_ARG, "refiner_input_plan: feature dtype must be GFN_F32 or GFN_F16");
    if (!f0 || !f1 || !flow || !d || (disp_dim > 0 && (!disp_w || !disp_b))) return gfn::fail(GFN_ERR_INVALID_ARG, "refiner_input_plan: null pointer");
    if (B < 0 || C <= 0 || Hs <= 0 || Ws <= 0 || G <= 0 || disp_dim < 0 || d_bs < (int64_t)(2 * C + disp_dim) * G * G || ((symmetric & 1) && (B & 1)) ||
        (symmetric & ~3) || (long)C * Hs * Ws >= (1L << 31) || B > 65535 || (long)G * G >= (1L << 31))
        return gfn::fail(GFN_ERR_INVALID_ARG, "refiner_input_plan: bad size");
    if (!lean_shape(C, Hs, Ws, G, r, dtype == GFN_F16))
        return gfn::fail(GFN_ERR_INVALID_ARG, "refiner_input_plan: the local correlation of this shape takes no plan (ask gfn_local_corr_plans first)");
    if (!scratch || scratch_bytes < gfn_local_corr_scratch_bytes(B, G) || ((uintptr_t)scratch & 3))
        return gfn::fail(GFN_ERR_SCRATCH, "refiner_input_plan: scratch too small");
    if (B == 0) return GFN_OK;
    gfn_ri::RiArgs q;
    q.fa = f0; q.fb = f1; q.flow = flow; q.dw = disp_w; q.db = disp_b; q.d = d; q.d_bs = (long)d_bs;
    q.B = B; q.Bh = (symmetric & 1) ? B / 2 : B; q.C = C; q.Hs = Hs; q.Ws = Ws; q.G = G; q.Dd = disp_dim; q.disp_scale = disp_scale;
    const bool keep = (symmetric & 2) != 0;  // the grid_feature planes are already in d
    LcParams p{};
    p.flow = flow; p.f16 = dtype == GFN_F16;
    p.B = B; p.C = C; p.G = G; p.H = Hs; p.W = Ws;
    p.todo = reinterpret_cast<int *>(scratch);
    p.plan = lean_plan_ptr(scratch, B, G);
    hipStream_t s = (hipStream_t)stream;
    const bool h = dtype == GFN_F16;
    switch (r) {
        case 1: return h ? launch_ri_plan<1, _Float16>(q, p, s, keep) : launch_ri_plan<1, float>(q, p, s, keep);
        case 2: return h ? launch_ri_plan<2, _Float16>(q, p, s, keep) : launch_ri_plan<2, float>(q, p, s, keep);
        case 3: return h ? launch_ri_plan<3, _Float16>(q, p, s, keep) : launch_ri_plan<3, float>(q, p, s, keep);
        case 4: return h ? launch_ri_plan<4, _Float16>(q, p, s, keep) : launch_ri_plan<4, float>(q, p, s, keep);
        case 5: return h ? launch_ri_plan<5, _Float16>(q, p, s, keep) : launch_ri_plan<5, float>(q, p, s, keep);
        case 6: return h ? launch_ri_plan<6, _Float16>(q, p, s, keep) : launch_ri_plan<6, float>(q, p, s, keep);
        default: return h ? launch_ri_plan<7, _Float16>(q, p, s, keep) : launch_ri_plan<7, float>(q, p, s, keep);
    }
}

GFN_EXPORT int gfn_local_corr_bwd_f0(const float *grad_out, int64_t grad_out_bs, const float *f1, const float *f1_second,
                                    const float *flow, float *grad_f0, int64_t grad_f0_bs, int B, int C, int G, int H, int W, int r,
                                    int grid_based, int win_h, int win_w, gfn_stream_t stream) {
    if (!grad_out || !f1 || !grad_f0) return gfn::fail(GFN_ERR_INVALID_ARG, "local_corr_bwd: null tensor pointer");
    if (f1_second && (B & 1)) return gfn::fail(GFN_ERR_INVALID_ARG, "local_corr_bwd: symmetric batch must be even");
    if (B < 0 || C <= 0 || G <= 0 || H <= 0 || W <= 0 || r < 0 || win_h <= 0 || win_w <= 0)
        return gfn::fail(GFN_ERR_INVALID_ARG, "local_corr_bwd: bad size B=%d C=%d G=%d H=%d W=%d r=%d", B, C, G, H, W, r);
    const long K = (long)(2 * r + 1) * (2 * r + 1);
    if (grad_f0_bs < (long)C * G * G || grad_out_bs < K * G * G)
        return gfn::fail(GFN_ERR_INVALID_ARG, "local_corr_bwd: batch stride smaller than one batch element");
    if (!flow && !(G == win_h && G == win_w))
        return gfn::fail(GFN_ERR_INVALID_ARG, "local_corr_bwd: flow=NULL needs num_grid == h == w");
    if ((long)C * H * W >= (1L << 31)) return gfn::fail(GFN_ERR_INVALID_ARG, "local_corr_bwd: tensor too large");
    if (B == 0) return GFN_OK;
    LcParams p{};
    p.f1 = f1; p.f1_second = f1_second; p.flow = flow;
    p.f16 = 0;
    p.Bh = f1_second ? B / 2 : B;
    p.B = B; p.C = C; p.G = G; p.H = H; p.W = W;
    p.sqrt_c = (float)sqrt((double)C);
    p.inv_sqrt_c = (float)(1.0 / sqrt((double)C));
    p.r = r; p.win_h = win_h; p.win_w = win_w; p.grid_based = grid_based;
    const long total = (long)B * ((C + kBwdCh - 1) / kBwdCh) * G * G;
    const unsigned grid = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(local_corr_bwd_f0_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, grad_out, (long)grad_out_bs, grad_f0,
                       (long)grad_f0_bs);
    return gfn::check_launch("local_corr_bwd_f0_kernel");
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
