// local_corr_mq.h -- round 3: large windows (r >= 5 on 64-channel maps: GFNet's r = 6 at stride 8 and r = 7 at stride 16) with the
// D-stage on the matrix core.  Included by local_corr.hip after local_corr_mstage.h: the split-bf16 formulation, the swizzled slots,
// the staging helpers (mm_item / mm_issue / mm_commit) and the guarded D buffer are that file's; what differs is the shape of the
// kernel around them.
//
// Why here and not at r = 3, 4: a cell at r = 6 / 7 takes 196 / 256 products per channel (r = 4: 100) and the round-1 kernel's
// D-stage feeds every v_fma_f32 with an LDS dword of its own -- at these radii that stage is ~2 k cycles per 16-channel chunk and
// wave, four chunks a tile.  On the matrix core the same chunk is <= 20 instructions of 16 cycles.  The persistent one-workgroup
// kernel of round 3 (deleted in round 5, see local_corr_mstage.h) lost that gain again at these shapes (two passes, 40 accumulator registers beside two passes' loads:
// spills; every phase in lockstep across the CU).  This kernel keeps the round-1 launch shape instead:
//   * a tile is 2 x 16 cells = two groups of 2 x 8 (the N = 16 of v_mfma_f32_16x16x32_bf16), one 8-wave workgroup per tile, two
//     workgroups per CU (80 KB of LDS each): one stages while the other multiplies;
//   * 16-channel chunks (64-byte slots: hi octets | lo octets, pieces swizzled as mm_swz<16>), four chunks a tile, the next
//     chunk's loads in flight across the current chunk's products; accumulators (<= 4 kMmNBW registers: four waves serve a group
//     as column tile x row parity) live across the chunks, the D buffer aliases the stage;
//   * no plan launch: the workgroup derives its region from the 32 flows itself (wave 0; the other waves' f0 loads cover the
//     round trip); a tile whose windows do not fit the groups' accumulators runs the round-1 routine (fp32 FMAs) in this workgroup,
//     which leaves what does not fit its stage either to the second launch (2 x 8-cell sub-tiles).
// Where a tile's ~27 k cycles go (tools/stamp_local_corr.py 64 64 70 40 6, GFN_ABLATE build): 4.5-5 k until the region is known (kernel
// arguments, the flows' round trip to memory, ~400 instructions of one wave), 6 k until chunk 0 is filed, 4 x 2.8 k per chunk -- of which
// the products are 0.3-0.6 k and the rest the next chunk's loads arriving and being split -- 4 k to file the accumulators and store.
// The matrix core took the D-stage off the critical path; what is left is the latency of dependent loads with two workgroups per CU to
// overlap it.  Tried without effect: the flows through the scalar cache (s_load_dwordx16 + v_writelane) and s_setprio for the set-up
// wave (the wait is the memory round trip, not queueing or issue slots); skipping empty staging items by a branch.
// Numerics: split-bf16 (local_corr_mstage.h; products exact in fp32 up to 2^-17 relative per term, fp32 accumulation; fp16 maps split
// exactly), not bit-identical to the fp32 FMA kernels.  -DGFN_MQ=0 builds keep r >= 5 on the round-1 kernel.

#ifndef GFN_MQ
#define GFN_MQ 1
#endif

constexpr int kMqLds = 80 * 1024;   // >= the round-1 routine's 68 KB stage + cells + f0 block at C = 64 (79 008 bytes)

template <int R, int C>
struct Mq {
    static constexpr int KC = 16, NCH = C / 16, NSUB = 1, NPIECE = 4, SLOT = 64;
    static constexpr int PW = 2 * R + 2, D = 2 * R + 1, K = D * D, TS = 2 * D + 1;
    static constexpr int NC = 32, NW = 8;
    static constexpr int RP = PW + 6;                           // D-buffer row: 3 guard floats | PW positions | 3 guard floats
    static constexpr int DS = ((PW * RP + 31) & ~31) + 5;
    static constexpr int NBW = kMmNBW;
    static constexpr int kCellBytes = (NC * 20 + 96 + 15) & ~15;   // five per-cell arrays + 24 header ints
    static constexpr int kTabBytes = (NC * TS * 4 + 15) & ~15;
    static constexpr int kF0Cell = NCH * 64 + 16;               // per cell: NCH x (hi 32 B | lo 32 B) + pad
    static constexpr int kF0Bytes = NC * kF0Cell;
    static constexpr int kDbufBytes = NC * DS * 4;
    static constexpr int kStage = (kMqLds - kCellBytes - kTabBytes - kF0Bytes) & ~127;
    static constexpr int kCap = kStage / SLOT - 32;             // positions that fit (a block may read 31 slots past the region's end)
    static_assert(kDbufBytes <= kStage, "the D buffer aliases the stage");
    // what process_tile<R, 1, true, kTileW, false> lays out when this kernel hands it a tile: stage + cell arrays + f0 block [32][C + 4]
    static_assert(kStageBytes + ((32 * 20 + 32 + 15) & ~15) + 32 * (C + 4) * 4 <= kMqLds, "the round-1 routine must fit this kernel's LDS");
    static_assert(C % 16 == 0 && C == 64, "built for 64-channel maps (8 waves x 8 channels of the f0 block)");
};

// cell ids: group g = id >> 4 holds columns 8 g .. 8 g + 7 of both tile rows (one DPP row of wave 0's lanes)
__device__ __forceinline__ int mq_cell_row(int c) { return (c >> 3) & 1; }
__device__ __forceinline__ int mq_cell_col(int c) { return ((c >> 4) << 3) | (c & 7); }
__device__ __forceinline__ int mq_cell_id(int r, int c) { return ((c >> 3) << 4) | (r << 3) | (c & 7); }

template <int R, int C, typename FT, bool QOK = true>
__global__ __launch_bounds__(kThreads, 4) void local_corr_mq_kernel(LcParams p) {
    typedef Mq<R, C> M;
    constexpr int PW = M::PW, D = M::D, K = M::K, TS = M::TS, NC = M::NC, RP = M::RP, DS = M::DS, NBW = M::NBW, NCH = M::NCH, NW = M::NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *dbuf = reinterpret_cast<float *>(smem);
    unsigned char *misc = smem + M::kStage;
    int *cellX0 = reinterpret_cast<int *>(misc);
    int *cellY0 = cellX0 + NC;
    float *cellNx = reinterpret_cast<float *>(cellY0 + NC);
    float *cellNy = cellNx + NC;
    int *cellFlag = reinterpret_cast<int *>(cellNy + NC);
    int *hdr = cellFlag + NC;   // [0..3] region x0, y0, w, h; [4] flagged cells; [5] path; [6] no window leaves the image; [8 + 4 g ..] group box x0, y0, y1
    float *tab = reinterpret_cast<float *>(misc + M::kCellBytes);
    unsigned char *f0b = misc + M::kCellBytes + M::kTabBytes;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tid = threadIdx.x;
    const int G = p.G, H = p.H, W = p.W;
    const float xhi = p.win_xhi, xlo = -xhi, yhi = p.win_yhi, ylo = -yhi;
    const unsigned GG4 = (unsigned)(G * G) * 4u;
    const unsigned wid = gfn::xcd_remap(blockIdx.x, gridDim.x);
    const int tiles = p.tiles_x * p.tiles_y;
    const int b = wid / tiles, tile = wid - b * tiles;
    const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    const int row0 = ty * 2, col0 = tx * kTileW;
#ifdef GFN_ABLATE
    const bool stamping = ABL(p, 512) && blockIdx.x == 2000 && (tid & 63) == 0 && (tid >> 6) < 2;
    long long stamp[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    STAMP(0);

    // ---- the f0 block: wave w takes channels 8 w .., lane = channel (lane >> 3) x quad of cells (tile row, four columns) ----------
    const int k8 = lane >> 3, fr = (lane & 7) >> 2, fc4 = (lane & 3) * 4;
    f32x4 f0v;
    {
        const bool in = (row0 + fr < G) & (col0 + fc4 < G);
        const rsrc_t f0r = make_rsrc(p.f0 + (size_t)b * p.f0_bs, (unsigned)C * GG4);
        const unsigned off = in ? (unsigned)((row0 + fr) * G + col0 + fc4) * 4u + (unsigned)(wave * 8 + k8) * GG4 : kOffRange;
        f0v = buf_ld4(f0r, off, 0u);
    }
    // ---- wave 0: the 32 flows -> cells, group boxes, the staging region ------------------------------------------------------------
    if (wave == 0) {  // scalar
        const int gi = row0 + mq_cell_row(lane), gj = col0 + mq_cell_col(lane);
        const bool ok = (lane < NC) & (gi < G) & (gj < G);
        const rsrc_t flr = make_rsrc(p.flow + (size_t)b * 2 * G * G, 2u * GG4);
        const unsigned fo = ok ? (unsigned)(gi * G + gj) * 4u : kOffRange;
        const float nx = buf_ld(flr, fo, 0u), ny = buf_ld(flr, fo, GG4);
        const CellBox c = cell_box<PW>(ok, ok ? nx : 0.f, ok ? ny : 0.f, xlo, ylo, W, H);
        if (lane < NC) {
            cellX0[lane] = c.X0;
            cellY0[lane] = c.Y0;
            cellNx[lane] = ok ? nx : 0.f;
            cellNy[lane] = ok ? ny : 0.f;
            cellFlag[lane] = c.flag;
        }
        const unsigned long long slow_mask = __ballot(c.flag == kCellSlow);
        // Windows are staged CLIPPED to the image (a 16-pixel window on the 32-pixel maps of stride 16 hangs over the border more often
        // than not): rows and columns outside it are neither staged nor multiplied; the D buffer of a border tile is zeroed before the
        // accumulators are filed.  (cell_box's "no window" markers +-kFar survive the clipping.)
        const int cx0 = max(c.bx0, 0), cy0 = max(c.by0, 0), cx1 = min(c.bx1, W), cy1 = min(c.by1, H);
        const int rx0 = row_min_i32(cx0), ry0 = row_min_i32(cy0), rx1 = -row_min_i32(-cx1), ry1 = -row_min_i32(-cy1);
        const bool grp_lane = (lane & 15) == 15 && lane < NC;
        if (grp_lane) {
            int *gb = hdr + 8 + (lane >> 4) * 4;
            gb[0] = rx0; gb[1] = ry0; gb[2] = ry1;
        }
        // a group's windows span at most two column tiles and 2 NBW rows of the region
        const bool grp_ok = !grp_lane || rx0 == kFar || ((rx1 - rx0 <= 32) & (ry1 - ry0 <= 2 * NBW));
        const bool groups_fit = __all(grp_ok);
        const int bx0 = wave_min_i32(cx0), by0 = wave_min_i32(cy0), bx1 = -wave_min_i32(-cx1), by1 = -wave_min_i32(-cy1);
        const bool all_in = __all(c.inside);
        if (lane == 0) {
            // the region starts on a multiple of 4 pixels where a quad could otherwise straddle the image's left edge (border tiles) and
            // where it is free (does not add a quad per row); fp16 maps: on an even pixel (8-byte quads at 4-byte alignment)
            int x0 = p.f16 ? (bx0 & ~1) : bx0;
            const int xa = bx0 & ~3;
            if (!all_in || ((W & 3) == 0 && ((bx1 - xa + 3) >> 2) == ((bx1 - bx0 + 3) >> 2))) x0 = xa;
            int w = max(bx1 - x0, 0), h = max(by1 - by0, 0), y0 = by0;
            if (w == 0 || h == 0) { x0 = 0; y0 = 0; w = 0; h = 0; }  // no window touches the image
            const int pitch = ((w + 3) >> 2) * 4;
            hdr[0] = x0; hdr[1] = y0; hdr[2] = w; hdr[3] = h;
            hdr[4] = __popcll(slow_mask);
            hdr[6] = all_in ? 1 : 0;
            // 1: this kernel's path; 2: the round-1 routine, here (the groups' accumulators do not cover the windows, but its stage
            // -- as many positions, no group constraint -- holds the region); 0: neither stage holds it: the second launch's list
            const bool fits = groups_fit && (long)pitch * h <= M::kCap && w <= 252 && h <= 255;
            // (the round-1 routine's own region: no start alignment, rows of whole quads)
            const int w4r = (max(bx1 - bx0, 0) + 3) & ~3;
            hdr[5] = fits ? 1 : ((long)w4r * h <= kStageBytes / (kSlotV4 * 16) - 1 ? 2 : 0);
        }
    }
    STAMP(1);
    __syncthreads();
    STAMP(2);
    MmRegion u;
    u.x0 = __builtin_amdgcn_readfirstlane(hdr[0]); u.y0 = __builtin_amdgcn_readfirstlane(hdr[1]);
    u.w = __builtin_amdgcn_readfirstlane(hdr[2]); u.h = __builtin_amdgcn_readfirstlane(hdr[3]);
    mm_region_geometry(u);
    const int path = __builtin_amdgcn_readfirstlane(hdr[5]);
    if (path != 1) {
        // A group's windows are spread over more than 32 columns or 2 NBW rows, or the region does not fit the stage (scattered or
        // strongly magnifying flow).  Where the round-1 routine's stage would hold the region (869 positions, no group
        // constraint) it takes the tile here and now (fp32 FMAs; on the raw soft-argmax flows of stride 16 a third of the tiles:
        // sent to the list they doubled the second launch, 47 -> 104 us at 448); the rest goes to the second launch's list.
        if (path == 2) {
            if (tid == 0 && (wid & 7u) == 0) atomicAdd(p.todo + 6, 8);  // informational, sampled (header word 6 -> 7: tiles the fp32 routine took here)
            __syncthreads();  // LDS is laid out anew
            process_tile<R, 1, true, kTileW, false, FT, 68 * 1024, QOK>(p, b, row0, col0, 2, wid, smem);
        } else if (tid == 0) {
            p.todo[kTodoHdr + atomicAdd(p.todo, 1)] = (int)wid;
        }
        return;
    }
    const int ipw = (((u.h * u.nq + 15) >> 4) + NW - 1) / NW;
    const rsrc_t f1r = make_rsrc(f1_of<FT>(p, b), (unsigned)C * (unsigned)(H * W) * (unsigned)sizeof(FT));
    const unsigned chunk_off = 16u * (unsigned)(H * W) * (unsigned)sizeof(FT);
    MmLane ml;
#pragma unroll
    for (int n = 0; n < kMmPre; ++n) ml.it[n] = mm_item<M, NW, true, FT, true>(u, H, W, wave, lane, n);
    MmRegs<FT> pre;
    mm_issue<true, FT, true>(pre, f1r, 0u, H, W, u, ipw, ml);
    STAMP(3);

    // ---- f0 block -> bf16 hi / lo: lanes k8 and k8 ^ 1 exchange, so that each files channel PAIRS (32-bit writes) of two cells -----
    {
        // (__builtin_bit_cast applied to a vector ELEMENT reads element 0 under hipcc 7.2: each element goes through a float of
        // its own first)
        float other[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float own = f0v[e];
            other[e] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, own), 0x128, 0xf, 0xf, false));  // row_ror:8 = lane ^ 8
        }
        const int odd = k8 & 1;
        const int ch = wave * 8 + (k8 & ~1);   // even channel of the pair
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
            const int e = 2 * odd + e2;        // this lane's two cells of the quad
            const int fc = fc4 + e;
            const bool fok = (row0 + fr < G) & (col0 + fc < G);
            const float mine = odd ? (e2 ? f0v[3] : f0v[2]) : (e2 ? f0v[1] : f0v[0]);
            const float theirs = odd ? (e2 ? other[3] : other[2]) : (e2 ? other[1] : other[0]);
            unsigned hi, lo;
            split_pair(fok ? (odd ? theirs : mine) : 0.f, fok ? (odd ? mine : theirs) : 0.f, hi, lo);
            unsigned *slot = reinterpret_cast<unsigned *>(f0b + mq_cell_id(fr, fc) * M::kF0Cell + (ch >> 4) * 64 + (ch & 15) * 2);
            slot[0] = hi;
            slot[8] = lo;
        }
    }
    // ---- fraction table: the reference's fp32 coordinate of every tap column / row of every cell ------------------------------------
    {
        const int cell = lane & 31;
        const float cnx = cellNx[cell], cny = cellNy[cell];
        const int cX0 = cellX0[cell], cY0 = cellY0[cell];
        bool tab_bad = false;
        constexpr int NTAB = (2 * D + 2 * NW - 1) / (2 * NW);
#pragma unroll
        for (int n = 0; n < NTAB; ++n) {
            const int a = 2 * wave + (lane >> 5) + n * 2 * NW;
            if (a < 2 * D) {
                const bool isy = a >= D;
                const int k = isy ? a - D : a;
                const float lin = isy ? gfn::linspace_step_at(ylo, yhi, p.win_ystep, D, k) : gfn::linspace_step_at(xlo, xhi, p.win_xstep, D, k);
                const float pix = unnorm((isy ? cny : cnx) + lin, isy ? H : W);
                const float fl = floorf(pix);
                const int origin = isy ? cY0 : cX0;
                tab_bad |= (origin != kFar) & !(fl == (float)(origin + k));
                tab[cell * TS + a] = pix - fl;
            }
        }
        if (tab_bad && atomicOr(&cellFlag[cell], kCellSlow) == 0) atomicAdd(&hdr[4], 1);  // rare
    }
    // this wave's part of its group: column tile mt, rows of parity rp of the box of the group's windows (region-relative)
    const int g = wave >> 2, mt = wave & 1, rp = (wave >> 1) & 1;  // scalars
    int gx0, gy0, nb;
    {
        const int bx0 = __builtin_amdgcn_readfirstlane(hdr[8 + 4 * g]), by0 = __builtin_amdgcn_readfirstlane(hdr[9 + 4 * g]),
                  by1 = __builtin_amdgcn_readfirstlane(hdr[10 + 4 * g]);
        const bool any = bx0 != kFar;
        gx0 = any ? bx0 - u.x0 + 16 * mt : 0;
        gy0 = any ? by0 - u.y0 + rp : 0;
        nb = any ? min((by1 - by0 - rp + 1) >> 1, NBW) : 0;
    }
    // lane = position m (lane & 15) of the block + 16 * k-group q: A = piece q of slot (row, gx0 + m) (q = 0, 1: the hi octets; 2, 3:
    // the lo octets).  Afterwards the lane holds, per block, four consecutive positions (columns gx0 + 4 q ..) of region row
    // gy0 + 2 i for cell g * 16 + m.
    const int mq = lane >> 4, mm = lane & 15;
    const unsigned slot0 = (unsigned)(gy0 * u.pitch + gx0 + mm);
    const unsigned sstep = (unsigned)(2 * u.pitch);
    const unsigned b_addr = (unsigned)((g * 16 + mm) * M::kF0Cell + (mq & 1) * 16);
    f32x4 acc[NBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    mm_commit<M, true, FT>(smem, pre, ipw, ml);
    mm_rest<M, NW, true, FT>(smem, f1r, 0u, H, W, u, ipw, wave, lane);
    STAMP(4);
    __syncthreads();
    STAMP(5);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const bool more = ch + 1 < NCH;
        const unsigned next_off = (unsigned)(ch + 1) * chunk_off;
        if (more) mm_issue<true, FT, true>(pre, f1r, next_off, H, W, u, ipw, ml);  // next chunk's loads: in flight across the products
        const bf16x8_t b1 = *reinterpret_cast<const bf16x8_t *>(f0b + b_addr + ch * 64);
        const bf16x8_t b2 = *reinterpret_cast<const bf16x8_t *>(f0b + b_addr + ch * 64 + 32);
        unsigned slot_c = slot0;
        asm volatile("" : "+v"(slot_c));   // operand addresses re-derived per chunk: kept across the chunks they cost NBW registers
        auto a_op = [&](int i) {
            const unsigned s = slot_c + (unsigned)i * sstep;
            return *reinterpret_cast<const bf16x8_t *>(smem + s * (unsigned)M::SLOT + (((unsigned)mq ^ mm_swz<16>(s)) << 4));
        };
#pragma unroll
        for (int i0 = 0; i0 < NBW; i0 += 2) {
            if (i0 < nb) {  // scalar.  A row past the group's box repeats its last row into values nobody files
                bf16x8_t a[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) a[j] = a_op(min(i0 + j, nb - 1));
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], b1, acc[i0 + j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], b2, acc[i0 + j], 0, 0, 0);
            }
        }
        if (more) {
            __syncthreads();  // everyone is done reading this chunk's pixels
            mm_commit<M, true, FT>(smem, pre, ipw, ml);
            mm_rest<M, NW, true, FT>(smem, f1r, next_off, H, W, u, ipw, wave, lane);
            __syncthreads();
        }
    }
    STAMP(6);
    __syncthreads();  // the D buffer aliases the stage
    if (__builtin_amdgcn_readfirstlane(hdr[6]) == 0) {  // border tile: window positions outside the image are zeros nobody computes
        float4 *d4 = reinterpret_cast<float4 *>(dbuf);
        for (int e = tid; e < NC * DS / 4; e += kThreads) d4[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
    }
    {
        const int cellg = g * 16 + mm;
        const int X0 = cellX0[cellg], Y0 = cellY0[cellg];
        const bool has = X0 != kFar;                                  // false: off the grid, flagged, or its window misses the image
        const int dx0 = has ? gx0 + 4 * mq - (X0 - u.x0) : -1000;    // window column of the first of the lane's four positions
        const int dy0 = has ? gy0 - (Y0 - u.y0) : 0;                 // window row of block 0
        const bool col_ok = (unsigned)(dx0 + 3) < (unsigned)(PW + 3);
        float *dwin = dbuf + cellg * DS + 3 + dy0 * RP + dx0;
        // a block's columns past the (clipped) region's row hold the next row's pixels: window positions right of the image are zeros
        const int pc = gx0 + 4 * mq;
        const bool c0 = pc < u.pitch, c1 = pc + 1 < u.pitch, c2 = pc + 2 < u.pitch, c3 = pc + 3 < u.pitch;
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
            if ((i & ~1) < nb) {  // scalar
                if (col_ok & ((unsigned)(dy0 + 2 * i) < (unsigned)PW) & (i < nb)) {
                    dwin[2 * i * RP + 0] = c0 ? acc[i][0] : 0.f;
                    dwin[2 * i * RP + 1] = c1 ? acc[i][1] : 0.f;
                    dwin[2 * i * RP + 2] = c2 ? acc[i][2] : 0.f;
                    dwin[2 * i * RP + 3] = c3 ? acc[i][3] : 0.f;
                }
            }
        }
    }
    STAMP(7);
    __syncthreads();
    STAMP(8);
    {
        // lanes 0-31 -> the tile's cells so that a wave stores whole 64-byte grid-row segments; tap row ky = 2 wave + (lane >> 5)
        const int er = (lane >> 4) & 1, ec = lane & 15;
        const int cell = mq_cell_id(er, ec);
        const int gi = row0 + er, gj = col0 + ec;
        const int flag = cellFlag[cell];
        if ((gi < G) & (gj < G) & !(flag & kCellSlow)) {
            const bool empty = (flag & kCellEmpty) != 0;
            const float *dc = dbuf + cell * DS + 3;
            const float *tc = tab + cell * TS;
            const unsigned goff = (unsigned)(gi * G + gj) * 4u;
            const rsrc_t outr = make_rsrc(p.out + (size_t)b * p.out_bs, (unsigned)K * GG4);
            constexpr int NR = (D + 2 * NW - 1) / (2 * NW);
#pragma unroll
            for (int n = 0; n < NR; ++n) {
                const int ky = 2 * wave + (lane >> 5) + n * 2 * NW;
                if (ky < D) {
                    const float wy1 = tc[D + ky];
                    const float wy1s = wy1 * p.inv_sqrt_c, wy0s = (1.f - wy1) * p.inv_sqrt_c;
                    const float *dd = dc + ky * RP;
                    float m[PW];
#pragma unroll
                    for (int x = 0; x < PW; ++x) m[x] = fmaf(dd[RP + x], wy1s, dd[x] * wy0s);
#pragma unroll
                    for (int kx = 0; kx < D; ++kx) {
                        const float wx1 = tc[kx];
                        const float val = fmaf(m[kx + 1], wx1, m[kx] * (1.f - wx1));
                        buf_st_nt(outr, goff + (unsigned)(ky * D + kx) * GG4, 0u, empty ? 0.f : val);
                    }
                }
            }
        }
    }
    STAMP(9);
    // ---- flagged cells: general per-tap routine (about one cell in 10^4) -----------------------------------------------------------
    const int nslow = __builtin_amdgcn_readfirstlane(hdr[4]);
    if (nslow != 0) {  // block-uniform, rare
        __syncthreads();
        if (tid == 0) {
            int n = 0;
            for (int cell = 0; cell < NC; ++cell)
                if ((cellFlag[cell] & kCellSlow) && (row0 + mq_cell_row(cell) < G) && (col0 + mq_cell_col(cell) < G)) cellX0[n++] = cell;
            hdr[4] = n;
            atomicAdd(p.todo + 4, n);  // informational (bench.py: flagged_cell_frac)
        }
        __syncthreads();
        const int totalk = hdr[4] * K;
        for (int e = tid; e < totalk; e += kThreads) {
            const int cell = cellX0[e / K], k = e % K;
            const int gi = row0 + mq_cell_row(cell), gj = col0 + mq_cell_col(cell);
            p.out[(size_t)b * p.out_bs + ((size_t)k * G + gi) * G + gj] =
                tap_general<FT>(p, b, gi, gj, k / D, k % D, D, cellNx[cell], cellNy[cell]);
        }
    }
#ifdef GFN_ABLATE
    if (stamping)
        printf("mq r%d wave %d (cycles): set-up %lld | barrier %lld | chunk 0 issued %lld | committed %lld | barrier %lld | products %lld | "
               "filed %lld | barrier %lld | stores issued %lld\n",
               R, tid >> 6, stamp[1] - stamp[0], stamp[2] - stamp[0], stamp[3] - stamp[0], stamp[4] - stamp[0], stamp[5] - stamp[0],
               stamp[6] - stamp[0], stamp[7] - stamp[0], stamp[8] - stamp[0], stamp[9] - stamp[0]);
#endif
}
