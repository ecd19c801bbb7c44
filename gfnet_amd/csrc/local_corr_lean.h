// local_corr_lean.h -- the round-2 tile path of the local correlation (r <= 4).  Included by local_corr.hip inside its
// anonymous namespace (shares LcParams, cell_coords, tap_general, lane_group, wave_min_i32, unnorm, f1_of).
//
// Same tile (4 x 16 cells), LDS slot layout, D-stage and arithmetic as process_tile<R, 2, true, 16, false>: results are
// bit-identical to the round-1 kernel (tests/test_local_corr_gpu.py).  What changed is everything around the D-stage.
// Round-1 evidence (profiles/local_corr_sq_pmc.json, s_memtime stamps in profiles/r02_local_corr_stamps.md): a tile lived
// ~33 k cycles of which the D-stage proper is 2 x 2.8 k; the rest were two serialised memory round trips in the set-up
// (flow -> bounding box -> barrier -> stage loads), the *issue* of ~550 narrow vector loads per tile (38 of 64 lanes
// active, ~13 cycles each through the address path), and 1397 vector instructions per wave for 448 FMAs, many of them
// quarter-rate integer multiplies and 64-bit address adds.
//   * plan launch: one wave per tile reads the flow, builds the tile's staging region and writes it to scratch (and the ids
//     of the tiles whose windows do not fit the stage straight to the second launch's list).  The tile kernel therefore
//     issues the plan load, its flow loads and the f0 block at entry, the stage loads as soon as the 16-byte plan is back:
//     one memory round trip and ONE barrier in front of the D-stage instead of two of each;
//   * staging loads are 16-byte quads along the image row (a work item = 64 / quads-per-row region rows x 4 channel planes:
//     four loads bring 4 pixels x 4 channels per lane, written as four 16-byte LDS slots): ~5x fewer vector-memory
//     instructions, every lane busy; the region starts on a multiple of 4 pixels so that the quads are 16-byte aligned;
//   * all global traffic goes through buffer instructions: descriptor (SGPRs) + scalar offset per plane / row group +
//     a per-lane offset that is constant for the tile -- no vector address arithmetic at all;
//   * the region is the UNclipped bounding box of the windows that touch the image, out-of-image pixels are staged as
//     zeros (border tiles only, a block-uniform variant), so the D-stage addresses need no per-position tests and no zero
//     slot; cells whose window misses the image entirely never enter the box and get exact zeros from the epilogue;
//   * tile = two 4 x 8-cell halves (cells 0-31 / 32-63), the epilogue still stores 64-byte row segments;
//   * fraction table: lane = cell, wave = tap index (no division); stores use scalar plane offsets.

constexpr int kMmNBW = 10;  // local_corr_mq.h: accumulator blocks per wave of the matrix-core kernel (a group's box: <= 2 kMmNBW rows)

#ifndef GFN_LEAN_STAGE2_KB
#define GFN_LEAN_STAGE2_KB 40
#endif
#ifndef GFN_LEAN_TAB_IN_STAGE
#define GFN_LEAN_TAB_IN_STAGE 1
#endif
#ifndef GFN_LEAN_STAGE4_KB
#define GFN_LEAN_STAGE4_KB 64
#endif
constexpr int kPlanInts = 16;  // per tile: region A x0, y0, (h << 16) | w, flags; region B x0, y0, (h << 16) | w, geometry of A;
                               // geometry of B, direction b, row0 | col0 << 16, spare ...  (geometry = pitch | quads per row << 8 | items << 16:
                               // round 4 -- the tile kernel no longer derives the regions' pitch and item count (region_fits) and the tile's
                               // direction / position (two integer divisions) in front of its first load: ~120 of a wave's ~1 550 instructions)
constexpr int kPlanV4 = kPlanInts / 4;
__device__ __forceinline__ int plan_geometry(const RowPlan &u) { return u.pitch | (u.nq << 8) | (u.nitems << 16); }
constexpr int kPlanInterior = 1, kPlanSecond = 2, kPlanHalves = 4;
// flags: kPlanInterior -- no staged pixel lies outside the image; kPlanSecond -- the tile is on the second launch's list;
// kPlanHalves -- the windows of the whole tile do not fit the stage, those of its two 4 x 8-cell halves do: region A serves
// cells 0-31, region B cells 32-63, staged one after the other (otherwise region A serves all 64 cells)

// cell id inside a tile -> (row, column): cells 0-31 are the left 4 x 8 half, 32-63 the right one
__device__ __forceinline__ int cell_row(int c) { return (c & 31) >> 3; }
__device__ __forceinline__ int cell_col(int c) { return (c & 7) + ((c >> 5) << 3); }

// stage bytes of the lean kernel: small windows (r <= 2: 6 x 6 patches over 16-channel maps) need half the region, and with
// 40 KB three workgroups fit a CU
template <int R>
struct Lean {
    // + cells, fraction table, f0 block <= 80 KB: two workgroups per CU.  r = 7: the D buffer that aliases the stage (64 cells x 257
    // floats) needs 65 792 bytes; with the f0 block staged chunk by chunk (kF0Chunk) the total stays at 80 160
    static constexpr int kDbuf = ((64 * ((2 * R + 2) * (2 * R + 2) + 1) + 16) * 4 + 15) & ~15;  // + 16: the skew of cells 32-63 (lean_tile)
    // Round 5 experiment (-DGFN_LEAN_STAGE4_KB=68): r = 3 / 4 stage the f0 block chunk by chunk too (5 KB instead of 9 KB on 32-channel
    // maps) and give the 4 KB to the stage, 870 positions instead of 819, so that the 7.4 % of the bench's scale-4 tiles whose regions
    // are 44 x 19 = 836 pixels are staged whole instead of as two halves.  Measured SLOWER (bench flows 105.8 vs 102.7 us, homography
    // flows 98.4 vs 95.7): the per-chunk f0 loads and filing cost every tile more than the halves cost the few.
    static constexpr int kStage34 = GFN_LEAN_STAGE4_KB * 1024;
    static constexpr int kStage = R <= 2 ? GFN_LEAN_STAGE2_KB * 1024 : (kDbuf > kStage34 ? kDbuf : kStage34);
    static constexpr bool kF0Chunk = R >= 5 || (R >= 3 && GFN_LEAN_STAGE4_KB > 64);  // 16 channels of the f0 block in LDS at a time
    // Round 5: r >= 3 keep the fraction table (64 cells x 19 floats) INSIDE the stage, behind the D buffer: it is filled after the main
    // loop (measured level with filling it up front, GFN_LEAN_TABLE_LATE above), when the stage holds nothing else, and its 5 KB go to
    // the stage instead -- 880 positions instead of 819 at r = 4: the 44 x 19-pixel regions of the bench's flows (7.4 % of the
    // scale-4 tiles) are staged whole instead of as two halves.  The workgroup's LDS total is unchanged.
    static constexpr bool kTabInStage = GFN_LEAN_TAB_IN_STAGE != 0 && R >= 3 && R <= 4;
    static constexpr int kTabBytes = ((64 * (2 * (2 * R + 1) + 1) + 16) * 4 + 15) & ~15;
    static constexpr int kStageLds = kStage + (kTabInStage ? kTabBytes : 0);   // bytes of LDS the stage spans
    static constexpr int kTabOff = (kDbuf + 15) & ~15;                         // table offset inside the stage (kTabInStage)
    static_assert(!kTabInStage || kTabOff + kTabBytes <= kStageLds, "the late table lies behind the D buffer");
    static constexpr int kCap = kStageLds / (kSlotV4 * 16);
    static constexpr int kMinWaves = (R <= 2 && GFN_LEAN_STAGE2_KB <= 44) ? 6 : 4;  // waves per SIMD the register allocation must allow
    static constexpr int PW = 2 * R + 2;
};

// what one cell asks of the stage: patch origin, flags, unclipped window (if it touches the image)
struct CellBox {
    int X0, Y0, flag;
    int bx0, by0, bx1, by1;
    bool inside;
};
constexpr int kCellSlow = 1, kCellEmpty = 2;

template <int PW>
__device__ __forceinline__ CellBox cell_box(bool ok, float nx, float ny, float xlo, float ylo, int W, int H) {
    CellBox c;
    c.X0 = kFar; c.Y0 = kFar; c.flag = 0;
    c.bx0 = kFar; c.by0 = kFar; c.bx1 = -kFar; c.by1 = -kFar;
    c.inside = true;
    if (ok) {
        // patch origin = floor of the reference's own fp32 coordinate of tap 0 (linspace(lo, hi, D)[0] == lo)
        const float fx = floorf(unnorm(nx + xlo, W));
        const float fy = floorf(unnorm(ny + ylo, H));
        if ((fx > -1e6f) & (fx < 1e6f) & (fy > -1e6f) & (fy < 1e6f)) {  // false for nan/inf
            const int ix = (int)fx, iy = (int)fy;
            if ((ix < W) & (ix + PW > 0) & (iy < H) & (iy + PW > 0)) {  // the window touches the image
                c.X0 = ix; c.Y0 = iy;
                c.bx0 = ix; c.by0 = iy; c.bx1 = ix + PW; c.by1 = iy + PW;  // unclipped: out-of-image pixels are staged as zeros
                c.inside = (ix >= 0) & (ix + PW <= W) & (iy >= 0) & (iy + PW <= H);
            } else {
                c.flag = kCellEmpty;
            }
        } else {
            c.flag = kCellSlow;  // non-finite / absurd flow: the per-tap routine decides
        }
    }
    return c;
}


template <int R>
__device__ __forceinline__ bool region_fits(RowPlan &u) {
    constexpr int PW = Lean<R>::PW;
    u.nq = (u.w + 3) >> 2;
    const int w4 = u.nq * 4;
    u.pitch = w4 + ((PW - w4) & 15);  // pitch == patch width (mod 16): conflict-free b128 reads across patch rows
    if ((long)u.pitch * u.h > Lean<R>::kCap && (long)w4 * u.h <= Lean<R>::kCap) u.pitch = w4;
    if ((long)u.pitch * u.h > Lean<R>::kCap || u.w > 64) return false;
    // a multiple of 8 items, so that they split evenly over the 8 waves (every wave issues the same number of loads: no
    // branches around loads, exact wait counts)
    u.nitems = ((u.h * u.nq + 15) / 16 + 7) & ~7;
    return true;
}

// min over each row of 16 lanes (a 2 x 8-cell group), valid in lane 15 of the row
__device__ __forceinline__ int row_min_i32(int v) {
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));  // row_shr:1
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));  // row_shr:2
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));  // row_shr:4
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));  // row_shr:8
    return v;
}

// ---- plan launch: a wave plans kPlanPerWave tiles (all their flow loads in flight together) -----------------------------
constexpr int kPlanPerWave = 4;
// one wave: the plans of tiles wid0 .. wid0 + kPlanPerWave - 1 (those below `total`)
template <int R>
__device__ __forceinline__ void plan_tiles(const LcParams &p, unsigned wid0, unsigned total) {
    constexpr int PW = Lean<R>::PW;
    const int lane = threadIdx.x & 63;
    const int tiles = p.tiles_x * p.tiles_y;
    if (wid0 >= total) return;
    float nx[kPlanPerWave], ny[kPlanPerWave];
    bool ok[kPlanPerWave];
#pragma unroll
    for (int t = 0; t < kPlanPerWave; ++t) {
        const unsigned wid = wid0 + t < total ? wid0 + t : total - 1;
        const int b = wid / tiles, tile = wid - b * tiles;
        const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
        const int gi = ty * 4 + cell_row(lane), gj = tx * kTileW + cell_col(lane);
        ok[t] = (gi < p.G) & (gj < p.G);
        const size_t o = ((size_t)b * 2 * p.G + (ok[t] ? gi : 0)) * p.G + (ok[t] ? gj : 0);
        nx[t] = p.flow[o];
        ny[t] = p.flow[o + (size_t)p.G * p.G];
    }
#pragma unroll
    for (int t = 0; t < kPlanPerWave; ++t) {
        const unsigned wid = wid0 + t;
        if (wid >= total) break;
        const CellBox c = cell_box<PW>(ok[t], nx[t], ny[t], -p.win_xhi, -p.win_yhi, p.W, p.H);
        // boxes of the four 2 x 8-cell groups (lanes 16 g .. 16 g + 15: one DPP row each), of the two halves (groups 0-1 / 2-3), and
        // of the tile (their union)
        const int rx0 = row_min_i32(c.bx0), ry0 = row_min_i32(c.by0), rx1 = row_min_i32(-c.bx1), ry1 = row_min_i32(-c.by1);
        int hx0[2], hy0[2], hx1[2], hy1[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int gx0[2], gy0[2], gx1[2], gy1[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int l15 = (2 * h + k) * 16 + 15;
                gx0[k] = __builtin_amdgcn_readlane(rx0, l15);
                gy0[k] = __builtin_amdgcn_readlane(ry0, l15);
                gx1[k] = -__builtin_amdgcn_readlane(rx1, l15);
                gy1[k] = -__builtin_amdgcn_readlane(ry1, l15);
            }
            hx0[h] = min(gx0[0], gx0[1]); hy0[h] = min(gy0[0], gy0[1]);
            hx1[h] = max(gx1[0], gx1[1]); hy1[h] = max(gy1[0], gy1[1]);
        }
        const bool all_in = __all(c.inside);
        if (lane == 0) {
            auto region = [&](int bx0, int by0, int bx1, int by1, RowPlan &u) {
                // Border tiles: the region starts on a multiple of 4 pixels, so that a quad never straddles the image's left
                // edge (a quad hanging over the right edge reads on into the next row, or past the tensor where the buffer
                // returns zeros, and is masked per pixel).  Interior tiles: the same alignment makes the quads 16-byte
                // aligned when rows are multiples of 4 pixels; taken where it is free, i.e. does not add a quad per row (an
                // extra quad pushes the row pitch to the next conflict-free value or the region out of the stage).
                // fp16 maps: a quad is 8 bytes and buffer loads want 4-byte alignment, so the region starts on an even pixel
                // (rows are an even number of pixels there: the host sends odd row lengths to the round-1 path).
                u.x0 = p.f16 ? (bx0 & ~1) : bx0;
                {
                    const int xa = bx0 & ~3;
                    if (!all_in || ((p.W & 3) == 0 && ((bx1 - xa + 3) >> 2) == ((bx1 - bx0 + 3) >> 2))) u.x0 = xa;
                }
                u.y0 = by0;
                u.w = max(bx1 - u.x0, 0);
                u.h = max(by1 - by0, 0);
                if (u.w == 0 || u.h == 0) { u.x0 = 0; u.y0 = 0; u.w = 0; u.h = 0; }  // no window touches the image
                return region_fits<R>(u);
            };
            RowPlan ua, ub;
            const bool border_ok = true;
            int flags = all_in ? kPlanInterior : 0;
            const bool full = region(min(hx0[0], hx0[1]), min(hy0[0], hy0[1]), max(hx1[0], hx1[1]), max(hy1[0], hy1[1]), ua) && border_ok;
            ub = ua;
            if (!full) {
                const bool fa = region(hx0[0], hy0[0], hx1[0], hy1[0], ua), fb = region(hx0[1], hy0[1], hx1[1], hy1[1], ub);
                flags |= (fa && fb && border_ok) ? kPlanHalves : kPlanSecond;
            }
            int4 pl0, pl1;
            pl0.x = ua.x0; pl0.y = ua.y0; pl0.z = (ua.h << 16) | ua.w; pl0.w = flags;
            pl1.x = ub.x0; pl1.y = ub.y0; pl1.z = (ub.h << 16) | ub.w; pl1.w = 0;
            pl1.w = plan_geometry(ua);
            const int wb = (int)(wid / (unsigned)tiles), wt = (int)(wid - (unsigned)wb * (unsigned)tiles);
            const int wty = wt / p.tiles_x, wtx = wt - wty * p.tiles_x;
            reinterpret_cast<int4 *>(p.plan)[kPlanV4 * wid] = pl0;
            reinterpret_cast<int4 *>(p.plan)[kPlanV4 * wid + 1] = pl1;
            reinterpret_cast<int4 *>(p.plan)[kPlanV4 * wid + 2] = make_int4(plan_geometry(ub), wb, (wty * 4) | ((wtx * kTileW) << 16), 0);
            if (flags & kPlanSecond) p.todo[kTodoHdr + atomicAdd(p.todo, 1)] = (int)wid;  // strong magnification / scattered flow: second launch
            else if ((flags & kPlanHalves) && (wid & 15u) == 0) atomicAdd(p.todo + 6, 16);  // informational, sampled: an atomic per tile
                                                                                            // on one word costs ~11 ns each
        }
    }
}

template <int R>
__global__ __launch_bounds__(256) void local_corr_plan_kernel(LcParams p) {
    plan_tiles<R>(p, (blockIdx.x * 4u + (threadIdx.x >> 6)) * kPlanPerWave, (unsigned)(p.B * p.tiles_x * p.tiles_y));
}

// The refiner-input kernel and the plan of the local correlation that follows it in ConvRefiner.forward (network.py:537-555) in
// ONE launch: blockIdx.x < q_blocks are refiner-input blocks of direction blockIdx.y, the rest plan that direction's tiles
// (16 per block).  Both only read the flow; the plan's ~8 us and a kernel boundary disappear from the local-correlation call.
template <int R, typename FT, bool KEEP>
__global__ __launch_bounds__(256) void refiner_input_plan_kernel(gfn_ri::RiArgs q, LcParams p, unsigned q_blocks, unsigned p_blocks, int banded) {
    const unsigned tiles = (unsigned)(p.tiles_x * p.tiles_y);
    if (banded) {  // 1-D grid: the refiner-input blocks of all directions in XCD-banded order (refiner_input.h), the plan blocks behind them
        const unsigned nq = 8u * gfn_ri::ri_band_slots(q_blocks) * (unsigned)q.B;
        if (blockIdx.x < nq) {
            int b;
            unsigned x;
            if (gfn_ri::ri_banded(blockIdx.x, q_blocks, q.Bh, b, x)) gfn_ri::refiner_input_cell<FT, KEEP>(q, b, x * 256u + threadIdx.x);
            return;
        }
        const unsigned pid = blockIdx.x - nq, pb = pid / p_blocks, px = pid - pb * p_blocks;
        const unsigned first = pb * tiles + (px * 4u + (threadIdx.x >> 6)) * kPlanPerWave;
        plan_tiles<R>(p, first, (pb + 1) * tiles);
        return;
    }
    const int b = gfn_ri::ri_direction(q.B, q.Bh, blockIdx.y);
    if (blockIdx.x < q_blocks) {
        gfn_ri::refiner_input_cell<FT, KEEP>(q, b, blockIdx.x * 256u + threadIdx.x);
        return;
    }
    const unsigned first = (unsigned)b * tiles + ((blockIdx.x - q_blocks) * 4u + (threadIdx.x >> 6)) * kPlanPerWave;
    plan_tiles<R>(p, first, (unsigned)(b + 1) * tiles);
}

// pp / PW for pp < 128 by a 24-bit multiply and a shift (the generic 32-bit magic division is a quarter-rate multiply-high)
template <int PW>
struct DivPW {
    static constexpr int S = 12;
    static constexpr int M = ((1 << S) + PW - 1) / PW;
    static constexpr bool exact() {
        for (int v = 0; v < 272; ++v)
            if (((v * M) >> S) != v / PW) return false;
        return true;
    }
    static_assert(exact(), "mul-shift division must be exact on the patch positions");
    __device__ static __forceinline__ int div(int v) { return (int)(((unsigned)v * (unsigned)M) >> S); }
};

// One tile.  CHECK: the region sticks out of the image (border tiles).  Everything that loads from global memory is
// straight-line code (no branches around loads), so that the compiler's wait counts stay exact: with conditional loads it
// falls back to vmcnt(0) and the cell set-up ends up waiting for the stage loads issued after it.
// HALVES: the tile is staged as two 4 x 8-cell halves, one after the other (region uA for cells 0-31 = round 0, uB for cells
// 32-63 = round 1); otherwise uA serves both rounds.
template <int R, int NCH, bool CHECK, bool HALVES, typename FT>
__device__ __forceinline__ void lean_tile(const LcParams &p, unsigned char *smem, const RowPlan &uA, const RowPlan &uB, int b, int row0, int col0,
                                          int tid, int lane, int wave) {
    constexpr int ROUNDS = 2;
    constexpr int C = 16 * NCH;
    constexpr int kStageBytes = Lean<R>::kStageLds;  // shadows the round-1 constant
    constexpr bool kTabIn = Lean<R>::kTabInStage;
    constexpr int PW = 2 * R + 2, P = PW * PW, NP = (P + 15) / 16;
    constexpr int D = 2 * R + 1, K = D * D;
    constexpr int NC = 64, DS = P + 1, TS = 2 * D + 1;
    // Round 5: the epilogue's lanes 0-31 hold cells c and c + 32 (lane -> cell so that a wave stores whole grid-row segments), whose D
    // and table rows start 32 DS / 32 TS dwords apart = on the same bank: every epilogue read was a 2-way conflict.  Rows of cells
    // 32-63 are skewed by 16 dwords.
    constexpr int kSkew = 16;
    constexpr bool F0CH = Lean<R>::kF0Chunk;       // the f0 block goes through LDS one 16-channel chunk at a time
    constexpr int CS = (F0CH ? kChunk : C) + 4;
    static_assert((NC * DS + kSkew) * 4 <= kStageBytes, "D buffer must fit in the stage it aliases");

    float4 *s4 = reinterpret_cast<float4 *>(smem);
    float *dbuf = reinterpret_cast<float *>(smem);
    int *cellX0 = reinterpret_cast<int *>(smem + kStageBytes);
    int *cellY0 = cellX0 + NC;
    float *cellNx = reinterpret_cast<float *>(cellY0 + NC);
    float *cellNy = cellNx + NC;
    int *cellFlag = reinterpret_cast<int *>(cellNy + NC);     // kCellSlow: redo per tap; kCellEmpty: window misses the image, result 0
    int *hdr = cellFlag + NC;                                  // [4]: number of flagged cells
    constexpr int kCellBytes = (NC * 20 + 32 + 15) & ~15;
    constexpr int kTabBytes = ((NC * TS + kSkew) * 4 + 15) & ~15;
    static_assert(kTabBytes == Lean<R>::kTabBytes, "table size");
    float *tab = reinterpret_cast<float *>(kTabIn ? smem + Lean<R>::kTabOff : smem + kStageBytes + kCellBytes);   // [NC][TS] per-tap fractions
    float *f0s = reinterpret_cast<float *>(smem + kStageBytes + kCellBytes + (kTabIn ? 0 : kTabBytes));  // [NC][C + 4]: the tile's f0, cell-major

    const int G = p.G, H = p.H, W = p.W;
    const float xhi = p.win_xhi, xlo = -xhi, yhi = p.win_yhi, ylo = -yhi;
    const unsigned GG4 = (unsigned)(G * G) * 4u;  // bytes of a (G,G) plane
#ifdef GFN_ABLATE
    const bool stamping = ABL(p, 512) && (blockIdx.x % 1999) == 1000 && (tid & 63) == 0 && (tid >> 6) < 2;
    long long stamp[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    STAMP(0);

    // ---- flow (wave 0), the f0 block and the first chunk's stage loads all go out at once ---------------------------------------
    // The tile kernels queue for the vector-memory pipe (a wave-load costs 30-40 cycles of the CU's texture-address path, and a
    // tile issued ~180 of them): every load that is not needed is gone.  lane = cell id (cells 0-31 left half, 32-63 right half);
    // only wave 0 reads the 64 flows and files the per-cell arrays; the fraction table is derived from those behind the barrier.
#ifndef GFN_LEAN_FLOW_ALL
#define GFN_LEAN_FLOW_ALL 0  // 1: every wave reads the flows (14 more loads per tile) and does its table share and patch addressing under the stage loads; 0: wave 0 only, both behind the barrier.  Measured level (r = 4: 98.1 / 148.1 vs 99.0 / 150.1 us): with two workgroups per CU the tile kernels are bound by the total of their vector instructions, not by where in the tile they sit
#endif
    constexpr bool kFlowAll = GFN_LEAN_FLOW_ALL != 0;
    const int my_gi = row0 + cell_row(lane), my_gj = col0 + cell_col(lane);
    const bool my_ok = (my_gi < G) & (my_gj < G);
    float my_nx = 0.f, my_ny = 0.f;
    if (kFlowAll || wave == 0) {  // scalar
        const rsrc_t flr = make_rsrc(p.flow + (size_t)b * 2 * G * G, 2u * GG4);
        const unsigned fo = my_ok ? (unsigned)(my_gi * G + my_gj) * 4u : 0u;
        my_nx = buf_ld(flr, fo, 0u);
        my_ny = buf_ld(flr, fo, GG4);
    }
    // the tile's f0 block: wave w takes channels w, w + 8, ...; one 16-byte load per lane = four consecutive cells of a grid row of
    // one of them (lane >> 4 = which channel, lane & 15 = tile row and column quad): C / 32 loads per wave instead of C / 8
    constexpr int NF0 = (F0CH ? kChunk : C) / kWaves;   // channels per wave (of the chunk, when staged chunk by chunk)
    constexpr int NF0L = (NF0 + 3) / 4;                 // loads per wave
    f32x4 f0q[NF0L];
    const int fk = lane >> 4, fr = (lane & 15) >> 2, fqd = lane & 3;
    const bool f0_lane = fk < (NF0 < 4 ? NF0 : 4);
    const rsrc_t f0r = make_rsrc(p.f0 + (size_t)b * p.f0_bs, (unsigned)C * GG4);
    // Round 5: where the f0 block is filed.  A wave's lanes hold (channel fk, tile row fr, column quad fqd); with rows of C + 4 floats
    // (16-byte aligned for the D-stage's float4 reads) tile row and column half do not reach the bank index, and with a wave's channels
    // 8 apart neither did the channel's low bits: the 32 lanes of a ds_write_b32 pass hit 4 banks (8-way conflict, 16 cycles per store
    // instead of 2).  Now a wave takes NF0 CONSECUTIVE channels and the row of cell c is c ^ (tile row of c) -- the tile row XORed into
    // the column's low bits: 16 banks per pass (2-way: free for 4-byte stores).  The D-stage reads row f0_row(cell).
    auto f0_row = [](int cell) { return cell ^ ((cell >> 3) & 3); };
    auto f0_issue = [&](int c0) {   // c0: first channel of the chunk (0 when the whole block is staged at once)
        const bool in = f0_lane & (row0 + fr < G) & (col0 + 4 * fqd < G);
#pragma unroll
        for (int l = 0; l < NF0L; ++l) {
            const int ch = c0 + wave * NF0 + 4 * l + fk;
            const unsigned fgoff = in ? (unsigned)((row0 + fr) * G + col0 + 4 * fqd) * 4u + (unsigned)ch * GG4 : 0u;
            f0q[l] = buf_ld4(f0r, fgoff, 0u);
        }
    };
    auto f0_commit = [&]() {
        if (f0_lane) {
#pragma unroll
            for (int l = 0; l < NF0L; ++l) {
                const int ch = wave * NF0 + 4 * l + fk;   // channel inside what f0s holds
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int fc = 4 * fqd + e;
                    const int fcell = ((fc >> 3) << 5) | (fr << 3) | (fc & 7);
                    const bool fok = (row0 + fr < G) & (col0 + fc < G);
                    f0s[f0_row(fcell) * CS + ch] = fok ? f0q[l][e] : 0.f;
                }
            }
        }
    };
    if (!ABL(p, 4)) f0_issue(0);
    // items of a chunk in flight per wave.  r <= 2: the 40 KB stage holds 512 positions = 512 lane items of a 16-channel chunk = one
    // item per lane, so a second register set only ever repeated the first item's loads -- half of the kernel's vector-memory
    // instructions, each a full trip through the texture-address path
    constexpr int PRE = (R <= 2 && Lean<R>::kCap <= 512) ? 1 : kQuadPre;
    auto quad_lane = [&](const RowPlan &u) {
        QuadLane ql;
#pragma unroll
        for (int n = 0; n < PRE; ++n) ql.it[n] = quad_item<CHECK, FT, kSlotV4, (R >= 3)>(u, H, W, wave, lane, n);
        return ql;
    };
    const QuadLane qlA = quad_lane(uA);
    const rsrc_t f1r = make_rsrc(f1_of<FT>(p, b), (unsigned)C * (unsigned)(H * W) * (unsigned)sizeof(FT));
    QuadRegs<PRE, FT> pre;
    if (!ABL(p, 1)) quad_issue<PRE, CHECK, FT, kSlotV4, (R >= 3)>(pre, f1r, 0u, H, W, uA, wave, lane, qlA, 0);
    STAMP(1);
    const QuadLane qlB = HALVES ? quad_lane(uB) : qlA;

    // ---- per-cell set-up (wave 0), f0 block and first chunk -> LDS ------------------------------------------------------------
    if (wave == 0) {  // scalar
        const CellBox c = cell_box<PW>(my_ok, my_ok ? my_nx : 0.f, my_ok ? my_ny : 0.f, xlo, ylo, W, H);
        cellX0[lane] = c.X0;
        cellY0[lane] = c.Y0;
        cellNx[lane] = my_ok ? my_nx : 0.f;
        cellNy[lane] = my_ok ? my_ny : 0.f;
        cellFlag[lane] = c.flag;
        const unsigned long long slow_mask = __ballot(c.flag == kCellSlow);
        if (lane == 0) hdr[4] = __popcll(slow_mask);
    }
    if (!ABL(p, 4)) f0_commit();
    auto fill_table = [&](float cnx, float cny, int cX0, int cY0) {
        // fraction table: the reference's fp32 coordinate of every tap column / row of every cell (local_correlation.py:55 adds
        // window offsets in normalised units, grid_sample un-normalises).  lane = cell, wave = tap index: no division.  Read by the
        // epilogue, barriers from here.
        bool tab_bad = false;
        constexpr int NTAB = (2 * D + kWaves - 1) / kWaves;
#pragma unroll
        for (int n = 0; n < NTAB; ++n) {
            const int a = wave + n * kWaves;   // scalar
            if (a < 2 * D) {
                const bool isy = a >= D;
                const int k = isy ? a - D : a;
                const float lin = isy ? gfn::linspace_step_at(ylo, yhi, p.win_ystep, D, k) : gfn::linspace_step_at(xlo, xhi, p.win_xstep, D, k);
                const float pix = unnorm((isy ? cny : cnx) + lin, isy ? H : W);
                const float fl = floorf(pix);
                const int origin = isy ? cY0 : cX0;
                // tap k must start at patch column/row k; if rounding moved its floor(), redo the cell per tap
                tab_bad |= (origin != kFar) & !(fl == (float)(origin + k));
                tab[lane * TS + (lane >> 5) * kSkew + a] = pix - fl;
            }
        }
        return tab_bad;
    };
    bool tab_bad = false;
    // ---- per-lane D-stage addressing: the float4 index of every (round, pass) patch position of the lane's cell ----------------------
    int g, s16;
    lane_group(lane, g, s16);
    const int cr = wave * 4 + g;  // cell inside a half (0..31)
    // Round 5: one register per (round, pass) holding the BYTE address of the position's slot (ds_read_b128 takes it as it is, the
    // four float4s of a slot through the instruction's offset field).  Rounds 2-4 packed two float4 indices per register because the
    // kernel sat at 128 VGPRs with the second-launch worker inlined; unpacking cost a v_and / v_bfe + v_lshl_add per position and
    // chunk (~65 of a wave's ~1 030 vector instructions at r = 4).  -DGFN_LEAN_APK=1: the packed form.
#ifndef GFN_LEAN_APK
#define GFN_LEAN_APK 0
#endif
    constexpr bool kApk = GFN_LEAN_APK != 0;
    unsigned apk[ROUNDS][kApk ? (NP + 1) / 2 : NP];
    auto addressing = [&](int rd, int X0, int Y0) {
        const RowPlan &u = (HALVES && rd == 1) ? uB : uA;
        // cells without a patch (off the grid, flagged, empty) read slot 0 onwards: valid memory, result unused
        const int base = X0 != kFar ? (Y0 - u.y0) * u.pitch + (X0 - u.x0) : 0;
#pragma unroll
        for (int t = 0; t < NP; ++t) {
            const int pp = s16 + 16 * t;
            const int yy = DivPW<PW>::div(pp), xx = pp - yy * PW;
            int slot = base + yy * u.pitch + xx;
            if (16 * t + 15 >= P) slot = pp < P ? slot : 0;  // positions past the patch (last pass only)
            const unsigned a = (unsigned)(slot * kSlotV4);
            if (!kApk)
                apk[rd][t] = a * 16u;
            else if (t & 1)
                apk[rd][t >> 1] |= a << 16;
            else
                apk[rd][t >> 1] = a;
        }
    };
    if (kFlowAll) {
        // every wave has the 64 flows (lane = cell id): its share of the fraction table AND its lanes' patch addresses (the origins
        // of the lane's two cells come from the lanes that hold them) are worked out under the stage loads' latency instead of
        // behind the barrier, where they were ~3 k of a tile's 22 k cycles at r = 4 with nothing in flight
        const CellBox c = cell_box<PW>(my_ok, my_ok ? my_nx : 0.f, my_ok ? my_ny : 0.f, xlo, ylo, W, H);
        tab_bad = fill_table(my_ok ? my_nx : 0.f, my_ok ? my_ny : 0.f, c.X0, c.Y0);
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) addressing(rd, __shfl(c.X0, rd * 32 + cr), __shfl(c.Y0, rd * 32 + cr));
    }
    STAMP(2);
    if (!ABL(p, 1)) {
        quad_commit<PRE, CHECK, FT>(s4, pre, H, W, uA, wave, lane, qlA, 0);
        quad_rest<CHECK, FT>(s4, f1r, 0u, H, W, uA, wave, lane, qlA, PRE);
    }
    STAMP(3);
    __syncthreads();
    STAMP(4);
    // Round 5 experiment: the fraction table is only read by the epilogue; filled behind the main loop (-DGFN_LEAN_TABLE_LATE=1) it is
    // out of the chain barrier -> addressing -> first product and runs beside the other waves' D-stages
#ifndef GFN_LEAN_TABLE_LATE
#define GFN_LEAN_TABLE_LATE 0  // measured level (r = 4: 96.5 vs 96.3 us, r = 2: 150.3 vs 153.9): the tile kernels are bound by what they issue, not by where in the tile it sits
#endif
    constexpr bool kTableLate = (GFN_LEAN_TABLE_LATE != 0 || kTabIn) && !kFlowAll;
    static_assert(!kTabIn || !kFlowAll, "the table in the stage is filled behind the main loop");
    if (!kTableLate) {
        if (!kFlowAll && !ABL(p, 32)) tab_bad = fill_table(cellNx[lane], cellNy[lane], cellX0[lane], cellY0[lane]);
        if (tab_bad && atomicOr(&cellFlag[lane], kCellSlow) == 0) atomicAdd(&hdr[4], 1);  // rare
    }

    if (!kFlowAll) {
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) addressing(rd, cellX0[rd * 32 + cr], cellY0[rd * 32 + cr]);
    }
    float acc[ROUNDS][NP];
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd)
#pragma unroll
        for (int t = 0; t < NP; ++t) acc[rd][t] = 0.f;
    STAMP(5);

    // ---- main loop: 16 channels at a time (per half when the tile is staged in halves) ------------------------------------
    constexpr int NS = HALVES ? 2 * NCH : NCH;  // steps: (half, chunk)
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        const int ch = HALVES ? st % NCH : st;          // compile-time after unrolling
        const int half = HALVES ? st / NCH : 0;
        const int c0 = ch * kChunk;
        const bool more = st + 1 < NS;
        const int nch = HALVES ? (st + 1) % NCH : st + 1, nhalf = HALVES ? (st + 1) / NCH : 0;
        const unsigned next_off = (unsigned)(nch * kChunk) * (unsigned)(H * W) * (unsigned)sizeof(FT);  // byte offset of the next step's first plane
        const RowPlan &un = (HALVES && nhalf == 1) ? uB : uA;
        const QuadLane &qn = (HALVES && nhalf == 1) ? qlB : qlA;
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd)
#pragma unroll
            for (int h = 0; h < (kApk ? (NP + 1) / 2 : NP); ++h) asm volatile("" : "+v"(apk[rd][h]));  // keep the (packed) addresses as they are: no recomputation per chunk
        if (more && !ABL(p, 1)) {  // next step's loads: in flight across this D-stage
            quad_issue<PRE, CHECK, FT, kSlotV4, (R >= 3)>(pre, f1r, next_off, H, W, un, wave, lane, qn, 0);
            if (F0CH) f0_issue(nch * kChunk);
        }
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            if (HALVES && rd != half) continue;
            if (ABL(p, 2)) continue;
            float f[kChunk];
            {
                const float4 *fq = reinterpret_cast<const float4 *>(f0s + f0_row(rd * 32 + cr) * CS + (F0CH ? 0 : c0));
                const float4 a0 = fq[0], a1 = fq[1], a2 = fq[2], a3 = fq[3];
                f[0] = a0.x; f[1] = a0.y; f[2] = a0.z; f[3] = a0.w; f[4] = a1.x; f[5] = a1.y; f[6] = a1.z; f[7] = a1.w;
                f[8] = a2.x; f[9] = a2.y; f[10] = a2.z; f[11] = a2.w; f[12] = a3.x; f[13] = a3.y; f[14] = a3.z; f[15] = a3.w;
            }
#pragma unroll
            for (int t = 0; t < NP; ++t) {
                const float4 *q = kApk ? s4 + ((t & 1) ? (apk[rd][t >> 1] >> 16) : (apk[rd][t >> 1] & 0xFFFFu))
                                       : reinterpret_cast<const float4 *>(smem + apk[rd][t]);
                const float4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3];
                float a = acc[rd][t];
                a = fmaf(f[0], v0.x, a);  a = fmaf(f[1], v0.y, a);  a = fmaf(f[2], v0.z, a);  a = fmaf(f[3], v0.w, a);
                a = fmaf(f[4], v1.x, a);  a = fmaf(f[5], v1.y, a);  a = fmaf(f[6], v1.z, a);  a = fmaf(f[7], v1.w, a);
                a = fmaf(f[8], v2.x, a);  a = fmaf(f[9], v2.y, a);  a = fmaf(f[10], v2.z, a); a = fmaf(f[11], v2.w, a);
                a = fmaf(f[12], v3.x, a); a = fmaf(f[13], v3.y, a); a = fmaf(f[14], v3.z, a); a = fmaf(f[15], v3.w, a);
                acc[rd][t] = a;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd)
#pragma unroll
            for (int t = 0; t < NP; ++t) asm volatile("" : "+v"(acc[rd][t]));  // pins the FMAs above this point
        STAMP(st == 0 ? 6 : 9);
        if (more) {
            __syncthreads();  // everyone is done reading this step's pixels
            STAMP(7);
            if (!ABL(p, 1)) {
                quad_commit<PRE, CHECK, FT>(s4, pre, H, W, un, wave, lane, qn, 0);
                quad_rest<CHECK, FT>(s4, f1r, next_off, H, W, un, wave, lane, qn, PRE);
            }
            if (F0CH) f0_commit();
            __syncthreads();
            STAMP(8);
        }
    }

    // ---- epilogue: D -> LDS, bilinear combination, coalesced stores ----------------------------------------------------
    if (kTableLate && !kTabIn) {
        if (!ABL(p, 32)) tab_bad = fill_table(cellNx[lane], cellNy[lane], cellX0[lane], cellY0[lane]);
        if (tab_bad && atomicOr(&cellFlag[lane], kCellSlow) == 0) atomicAdd(&hdr[4], 1);  // rare
    }
    __syncthreads();
    STAMP(10);
    if (kTabIn) {   // behind the barrier: nobody reads staged pixels any more, the table goes where they were (beside the D buffer)
        if (!ABL(p, 32)) tab_bad = fill_table(cellNx[lane], cellNy[lane], cellX0[lane], cellY0[lane]);
        if (tab_bad && atomicOr(&cellFlag[lane], kCellSlow) == 0) atomicAdd(&hdr[4], 1);  // rare
    }
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
        const int cell = rd * 32 + cr;
#pragma unroll
        for (int t = 0; t < NP; ++t) {
            const int pp = s16 + 16 * t;
            if (pp < P && !ABL(p, 16)) dbuf[cell * DS + rd * kSkew + pp] = acc[rd][t];
        }
    }
    STAMP(11);
    __syncthreads();
    STAMP(12);
    {
        // lane -> cell so that a wave stores whole 64-byte grid-row segments: tile row lane >> 4, column lane & 15
        const int er = lane >> 4, ec = lane & 15;
        const int cell = ((ec >> 3) << 5) | (er << 3) | (ec & 7);
        const int gi = row0 + er, gj = col0 + ec;
        const int flag = cellFlag[cell];
        if ((gi < G) & (gj < G) & !(flag & kCellSlow) & !ABL(p, 8)) {
            const bool empty = (flag & kCellEmpty) != 0;
            const float *dc = dbuf + cell * DS + (ec >> 3) * kSkew;
            const float *tc = tab + cell * TS + (ec >> 3) * kSkew;
            const unsigned goff = (unsigned)(gi * G + gj) * 4u;
            const rsrc_t outr = make_rsrc(p.out + (size_t)b * p.out_bs, (unsigned)K * GG4);
            float wx1[D], wx0[D];
#pragma unroll
            for (int kx = 0; kx < D; ++kx) { wx1[kx] = tc[kx]; wx0[kx] = 1.f - wx1[kx]; }
            constexpr int NR = (D + kWaves - 1) / kWaves;
#pragma unroll
            for (int n = 0; n < NR; ++n) {
                const int ky = wave + n * kWaves;  // scalar
                if (ky < D) {
                    // separable bilinear: the PW patch columns are blended vertically once (1/sqrt(C) folded into the row
                    // weights), every tap is then two instructions
                    const float wy1 = tc[D + ky];
                    const float wy1s = wy1 * p.inv_sqrt_c, wy0s = (1.f - wy1) * p.inv_sqrt_c;
                    const float *d = dc + ky * PW;
                    float m[PW];
#pragma unroll
                    for (int x = 0; x < PW; ++x) m[x] = fmaf(d[PW + x], wy1s, d[x] * wy0s);
#pragma unroll
                    for (int kx = 0; kx < D; ++kx) {
                        const float val = fmaf(m[kx + 1], wx1[kx], m[kx] * wx0[kx]);
                        if (!ABL(p, 128)) buf_st_nt(outr, goff, (unsigned)(ky * D + kx) * GG4, empty ? 0.f : val);
                        else asm volatile("" :: "v"(val));  // streamed: nothing on the hot path reads it back
                    }
                }
            }
        }
    }
    STAMP(13);
#ifdef GFN_ABLATE
    if (stamping)
        printf("lean r%d wave %d (cycles from entry): all issued %lld | cells+table+f0 in LDS %lld | stage0 committed %lld | barrier %lld | addressing %lld | "
               "D0 %lld | barrier %lld | stage1 committed+barrier %lld | D1 %lld | barrier %lld | dbuf %lld | barrier %lld | stores issued %lld\n",
               R, tid >> 6, stamp[1] - stamp[0], stamp[2] - stamp[0], stamp[3] - stamp[0], stamp[4] - stamp[0], stamp[5] - stamp[0], stamp[6] - stamp[0],
               stamp[7] - stamp[0], stamp[8] - stamp[0], stamp[9] - stamp[0], stamp[10] - stamp[0], stamp[11] - stamp[0], stamp[12] - stamp[0],
               stamp[13] - stamp[0]);
#endif

    // ---- flagged cells: general per-tap routine (about one cell in 10^4) ---------------------------------------------
#ifdef GFN_LEAN_ANALYZE
    return;
#endif
    const int nslow = __builtin_amdgcn_readfirstlane(hdr[4]);
    if (nslow != 0) {  // block-uniform, rare
        __syncthreads();
        if (tid == 0) {
            int n = 0;
            for (int cell = 0; cell < NC; ++cell)
                if ((cellFlag[cell] & kCellSlow) && (row0 + cell_row(cell) < G) && (col0 + cell_col(cell) < G)) cellX0[n++] = cell;
            hdr[4] = n;
            atomicAdd(p.todo + 4, n);  // informational (bench.py: flagged_cell_frac)
        }
        __syncthreads();
        const int total = hdr[4] * K;
        for (int e = tid; e < total; e += kThreads) {
            const int cell = cellX0[e / K], k = e % K;
            const int gi = row0 + cell_row(cell), gj = col0 + cell_col(cell);
            p.out[(size_t)b * p.out_bs + ((size_t)k * G + gi) * G + gj] =
                tap_general<FT>(p, b, gi, gj, k / D, k % D, D, cellNx[cell], cellNy[cell]);
        }
    }
}

// The first kLeanWorkers workgroups of the launch (the first to be dispatched) are the "second launch": they finish the
// tiles the plan launch listed as not fitting the stage even in halves, with the round-1 sub-tile routine, while the other
// workgroups stage their tiles -- no separate launch (5 us when its list is empty, ~14 us of serial tail when it holds a
// handful of tiles) and no tail.
// Only where the register budget allows it (r >= 3: 128 VGPRs per lane; the r <= 2 kernels live on 80 for three workgroups per
// CU and keep the separate second launch).  One worker per CU: with a handful of listed tiles all but a few leave at once,
// under wild flow (every tile listed, the tile workgroups leaving at once) they have the chip like the separate launch had.
// Round 5 A/B: with the worker inlined the r >= 3 kernels' metadata shows 72-79 VGPR spills, ~160 SGPR spills and
// ~300 bytes of scratch -- all of it in the worker branch (the ISA of the four lean_tile variants has no scratch access and 20 SGPR
// spills, exactly what the -DGFN_LEAN_INLINE_WORKERS=0 build shows for the whole kernel: 117-123 VGPRs, no spills, no scratch).
// As a launch of its own behind the tile kernel the worker costs nothing when the list is empty (94.7 vs 95.1 us, r = 4, 64
// directions) but 12 us of serial tail when three tiles are listed, which is the bench's situation (95.9 vs 108.5 us): the inlined
// form stays the default.
#ifndef GFN_LEAN_INLINE_WORKERS
#define GFN_LEAN_INLINE_WORKERS 1
#endif
template <int R>
constexpr int lean_workers() { return (GFN_LEAN_INLINE_WORKERS && R >= 3) ? 256 : 0; }  // a multiple of 8: the XCD of a tile's workgroup does not change

// one 4 x 16-cell tile `wid` from its plan record: the body of the tile kernel
template <int R, int NCH, typename FT>
__device__ __forceinline__ void lean_small_tile(const LcParams &p, unsigned char *smem, unsigned wid, int tid, int lane, int wave) {
    // the tile's plan through the scalar cache: a vector load of it would queue behind whatever the CU's other workgroups
    // have in the vector-memory pipeline
    typedef int i32x8 __attribute__((ext_vector_type(8)));
    typedef int i32x4s __attribute__((ext_vector_type(4)));
    i32x8 pl;
    i32x4s pg;
    {
        const int *pp = p.plan + (size_t)wid * kPlanInts;
        asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx4 %1, %2, 0x20\n\ts_waitcnt lgkmcnt(0)" : "=&s"(pl), "=&s"(pg) : "s"(pp) : "memory");
    }
    const int flags = pl[3];
    if (flags & kPlanSecond) return;  // the plan launch has put this tile on the second launch's list (block-uniform)
    RowPlan uA, uB;
    uA.x0 = pl[0]; uA.y0 = pl[1]; uA.w = pl[2] & 0xffff; uA.h = pl[2] >> 16;
    uB.x0 = pl[4]; uB.y0 = pl[5]; uB.w = pl[6] & 0xffff; uB.h = pl[6] >> 16;
    // pitch, quads per row and items as the plan launch worked them out (region_fits, checked there)
    uA.pitch = pl[7] & 0xff; uA.nq = (pl[7] >> 8) & 0xff; uA.nitems = pl[7] >> 16;
    uB.pitch = pg[0] & 0xff; uB.nq = (pg[0] >> 8) & 0xff; uB.nitems = pg[0] >> 16;
    if (ABL(p, 64)) {  // timing experiment: what a row ring would stage per tile
        uA.h = min(uA.h, 7); uB.h = min(uB.h, 7);
        (void)region_fits<R>(uA);
        (void)region_fits<R>(uB);
    }
    const int b = pg[1], row0 = pg[2] & 0xffff, col0 = pg[2] >> 16;
    const bool interior = (flags & kPlanInterior) != 0;
#ifdef GFN_LEAN_ANALYZE  // ISA reading aid (tools/dump_isa.py): only the interior whole-tile variant, so that the hot path is straight-line code in the dump
    lean_tile<R, NCH, false, false, FT>(p, smem, uA, uB, b, row0, col0, tid, lane, wave);
    return;
#endif
    if (flags & kPlanHalves) {
        if (interior) lean_tile<R, NCH, false, true, FT>(p, smem, uA, uB, b, row0, col0, tid, lane, wave);
        else lean_tile<R, NCH, true, true, FT>(p, smem, uA, uB, b, row0, col0, tid, lane, wave);
    } else {
        if (interior) lean_tile<R, NCH, false, false, FT>(p, smem, uA, uB, b, row0, col0, tid, lane, wave);
        else lean_tile<R, NCH, true, false, FT>(p, smem, uA, uB, b, row0, col0, tid, lane, wave);
    }
}

template <int R, int NCH, typename FT>
__global__ __launch_bounds__(kThreads, Lean<R>::kMinWaves) void local_corr_tile2_kernel(LcParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kLeanWorkers = lean_workers<R>();
    if constexpr (kLeanWorkers > 0) {
        if (blockIdx.x < kLeanWorkers) {  // block-uniform
            second_launch_worker<R, 2, FT, Lean<R>::kStage>(p, smem, (int)blockIdx.x, kLeanWorkers);  // (its own LDS layout inside the same allocation)
            return;
        }
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned wid = gfn::xcd_remap(blockIdx.x - kLeanWorkers, gridDim.x - kLeanWorkers);
    lean_small_tile<R, NCH, FT>(p, smem, wid, tid, lane, wave);
}
