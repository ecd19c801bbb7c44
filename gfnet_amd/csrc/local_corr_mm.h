// local_corr_mm.h -- round 3: the D-stage of the lean local-correlation tile on the matrix core.  Included by local_corr.hip
// after local_corr_lean.h (same namespace: plan, staging loads, fraction table and epilogue are the lean path's).
//
// Why (profiles/local_corr_sq_pmc.json, round 2): at r = 4 a wave issued 949 vector instructions of which 448 were the D-stage's
// v_fma_f32, every one of them fed by its own LDS dword (4 ds_read_b128 per 16 FMAs): the FMA issue and the LDS reads of the
// D-stage were each worth a third of the kernel, and they ran one after the other behind barriers.  v_pk_fma_f32 issues at half
// rate on gfx950, so the products have to leave the VALU.
//
// How: D[cell][position] = sum_c f0[cell][c] * f1[c][position] is a (positions x channels) . (channels x cells) product.  Per
// group of 2 x 8 cells (one DPP row of the tile's 64 lanes, 16 cells = the N of v_mfma_f32_16x16x32_bf16) the positions any of
// its windows touches are a box of at most 32 columns x NBW rows of the staged region; a block = 16 consecutive positions of
// one region row (M) x the 16 cells, and two waves serve a group (one per 16-column tile).  fp32 accuracy from bf16 operands:
// every value is split x = hi + lo (both round-to-nearest bf16, residual <= 2^-18 |x|) when it is filed in LDS, and the K = 32
// of the instruction holds a 16-channel chunk as [hi | lo]:
//      A (positions) = [f1_hi(16) | f1_lo(16)],  B1 (cells) = [f0_hi | f0_hi],  B2 = [f0_lo | f0_lo]
//      mfma(A, B1) + mfma(A, B2) = sum_c (f1_hi + f1_lo) (f0_hi + f0_lo)      -- all four partial products, fp32 accumulation
// so a product is wrong by at most 2 * 2^-18 relative (tests: 1e-4 * max(1, |ref|) against the oracle; measured ~2e-6).
// A position's slot keeps the 80 bytes of the fp32 stage (hi 32 B | lo 32 B | 16 B pad: the plan's capacity rule is unchanged and
// 16 consecutive slots start in 16 different 4-bank groups), so the A operand of a block is ONE ds_read_b128 per lane
// (lane = position + 16 * k-group) serving both instructions.  About a quarter of the products land inside some cell's
// window; the matrix core does 16 x the VALU's rate, and the LDS feeds it one dword per 64 products instead of one per product.
// The accumulators stay in registers across the channel chunks (4 per block); afterwards each lane holds four consecutive
// positions of one cell and writes those inside the cell's window to the D buffer (rows of PW + 6 floats: three guard columns on
// either side take the positions that hang over, so a lane's four values need one range test), which the lean epilogue reads.
// fp16 feature maps are split the same way (exactly: 11 significant bits fit hi + lo).
//
// Numerics class: NOT bit-identical to the fp32 FMA kernels (variant 4: the round-2 lean kernel, variant 2: round 1), which stay
// as cross-checks; the plan sends a tile whose group boxes exceed 32 x NBW to the second launch's list.

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

// (a, b) -> packed bf16 pairs (hi, lo) with a = hi.x + lo.x + O(2^-18 a)
__device__ __forceinline__ void split_pair(float a, float b, unsigned &hi, unsigned &lo) {
    const f32x2_t v = {a, b};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
    const f32x2_t h = {__builtin_bit_cast(float, hi << 16), __builtin_bit_cast(float, hi & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(v - h, bf16x2_t));
}

constexpr int kSlot8 = 2 * kSlotV4;  // 8-byte pieces per staged position (80-byte slot)

// commit of a staging work item into the bf16 stage: the lane's 4 pixels x 4 channels (channel quad cg of the chunk) become,
// per pixel, 8 bytes of the slot's hi half and 8 bytes of its lo half (piece cg of each)
template <int N, bool CHECK, typename FT>
__device__ __forceinline__ void quad_commit_mm(unsigned char *stage, const QuadRegs<N, FT> &r, int H, int W, const RowPlan &u, int wave, int lane,
                                               const QuadLane &ql, int k0) {
    const int ipw = u.nitems >> 3;
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const unsigned meta = (k0 == 0 && n < kQuadPre) ? ql.it[n].meta : quad_item<CHECK, FT, kSlot8>(u, H, W, wave, lane, k0 + n).meta;
        if ((k0 + n < ipw) & ((meta >> 17) & 1u)) {
            u32x2_t *dst = reinterpret_cast<u32x2_t *>(stage) + (meta & 0x1FFFu);
            unsigned m = 0xFu;
            if (CHECK) m = ((meta >> 18) & 1u) ? (meta >> 13) & 0xFu : 0u;
            const f32x4 w0 = QuadRaw<FT>::widen(r.a[n][0]), w1 = QuadRaw<FT>::widen(r.a[n][1]), w2 = QuadRaw<FT>::widen(r.a[n][2]),
                        w3 = QuadRaw<FT>::widen(r.a[n][3]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool in = !CHECK || ((m >> k) & 1u);
                unsigned h01, l01, h23, l23;
                split_pair(in ? w0[k] : 0.f, in ? w1[k] : 0.f, h01, l01);
                split_pair(in ? w2[k] : 0.f, in ? w3[k] : 0.f, h23, l23);
                dst[k * kSlot8] = u32x2_t{h01, h23};
                dst[k * kSlot8 + 4] = u32x2_t{l01, l23};
            }
        }
    }
}

template <bool CHECK, typename FT>
__device__ __forceinline__ void quad_rest_mm(unsigned char *stage, rsrc_t f1r, unsigned chunk_off, int H, int W, const RowPlan &u, int wave,
                                             int lane, const QuadLane &ql, int done) {
    for (int k0 = done; k0 < (u.nitems >> 3); ++k0) {  // only regions of more than 256 quads (rare)
        QuadRegs<1, FT> r;
        quad_issue<1, CHECK, FT, kSlot8>(r, f1r, chunk_off, H, W, u, wave, lane, ql, k0);
        quad_commit_mm<1, CHECK, FT>(stage, r, H, W, u, wave, lane, ql, k0);
    }
}

// One tile on the matrix core.  Template parameters and the set-up as lean_tile (local_corr_lean.h); cell ids are the lean
// path's: id = half * 32 + row * 8 + column-in-half, so that group g = id >> 4 is rows 2 (g & 1), 2 (g & 1) + 1 of half g >> 1.
template <int R, int NCH, bool CHECK, bool HALVES, typename FT>
__device__ __forceinline__ void lean_tile_mm(const LcParams &p, unsigned char *smem, const RowPlan &uA, const RowPlan &uB, unsigned wid, int tid,
                                             int lane, int wave) {
    constexpr int C = 16 * NCH;
    constexpr int kStageBytes = Lean<R>::kStage;
    constexpr int PW = 2 * R + 2;
    constexpr int D = 2 * R + 1, K = D * D;
    constexpr int NC = 64, TS = 2 * D + 1;
    constexpr int RP = PW + 6;           // D-buffer row: 3 guard floats | PW positions | 3 guard floats
    constexpr int DS = PW * RP + 1;      // odd: the 16 cells of a group start in different banks
    constexpr int NBW = Lean<R>::NBW;
    static_assert(NC * DS * 4 <= kStageBytes, "D buffer must fit in the stage it aliases");

    float *dbuf = reinterpret_cast<float *>(smem);
    int *cellX0 = reinterpret_cast<int *>(smem + kStageBytes);
    int *cellY0 = cellX0 + NC;
    float *cellNx = reinterpret_cast<float *>(cellY0 + NC);
    float *cellNy = cellNx + NC;
    int *cellFlag = reinterpret_cast<int *>(cellNy + NC);
    int *hdr = cellFlag + NC;
    constexpr int kCellBytes = (NC * 20 + 32 + 15) & ~15;
    constexpr int kTabBytes = (NC * TS * 4 + 15) & ~15;
    float *tab = reinterpret_cast<float *>(smem + kStageBytes + kCellBytes);
    // the tile's f0 as B operands: per cell NCH slots of 64 bytes = hi(16 channels) | lo(16 channels), bf16, + 16 bytes of pad
    // (cell stride = 4 (mod 16) dwords: the 16 cells of a group's operand read start in 16 different 4-bank groups)
    constexpr int kF0Cell = NCH * 64 + 16;
    static_assert(kF0Cell <= (C + 4) * 4, "the B operands take the place of the lean kernel's fp32 f0 block");
    unsigned char *f0b = smem + kStageBytes + kCellBytes + kTabBytes;

    const int G = p.G, H = p.H, W = p.W;
    const int tiles = p.tiles_x * p.tiles_y;
    const int b = wid / tiles, tile = wid - b * tiles;
    const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    const int row0 = ty * 4, col0 = tx * kTileW;
    const float xhi = p.win_xhi, xlo = -xhi, yhi = p.win_yhi, ylo = -yhi;
    const unsigned GG4 = (unsigned)(G * G) * 4u;
#ifdef GFN_ABLATE
    const bool stamping = ABL(p, 512) && (blockIdx.x % 1999) == 1000 && (tid & 63) == 0 && (tid >> 6) < 2;
    long long stamp[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    STAMP(0);

    // ---- flow, the f0 block and the first chunk's stage loads all go out at once (as lean_tile) -----------------------------
    const int my_gi = row0 + cell_row(lane), my_gj = col0 + cell_col(lane);
    const bool my_ok = (my_gi < G) & (my_gj < G);
    float my_nx, my_ny;
    {
        const rsrc_t flr = make_rsrc(p.flow + (size_t)b * 2 * G * G, 2u * GG4);
        const unsigned fo = my_ok ? (unsigned)(my_gi * G + my_gj) * 4u : 0u;
        my_nx = buf_ld(flr, fo, 0u);
        my_ny = buf_ld(flr, fo, GG4);
    }
    constexpr int NF0 = C / kWaves;
    float f0v[NF0];
    const int fr = lane >> 4, fc = lane & 15;
    const bool fok = (row0 + fr < G) & (col0 + fc < G);
    {
        const unsigned fgoff = fok ? (unsigned)((row0 + fr) * G + col0 + fc) * 4u : 0u;
        const rsrc_t f0r = make_rsrc(p.f0 + (size_t)b * p.f0_bs, (unsigned)C * GG4);
#pragma unroll
        for (int k = 0; k < NF0; ++k) f0v[k] = buf_ld(f0r, fgoff, (unsigned)(wave * NF0 + k) * GG4);  // NF0 consecutive channels
    }
    auto quad_lane = [&](const RowPlan &u) {
        QuadLane ql;
#pragma unroll
        for (int n = 0; n < kQuadPre; ++n) ql.it[n] = quad_item<CHECK, FT, kSlot8>(u, H, W, wave, lane, n);
        return ql;
    };
    const QuadLane qlA = quad_lane(uA);
    const rsrc_t f1r = make_rsrc(f1_of<FT>(p, b), (unsigned)C * (unsigned)(H * W) * (unsigned)sizeof(FT));
    constexpr int PRE = kQuadPre;
    QuadRegs<PRE, FT> pre;
    quad_issue<PRE, CHECK, FT, kSlot8>(pre, f1r, 0u, H, W, uA, wave, lane, qlA, 0);
    STAMP(1);
    const QuadLane qlB = HALVES ? quad_lane(uB) : qlA;

    // ---- per-cell set-up, fraction table, f0 block and first chunk -> LDS ----------------------------------------------------
    const CellBox c = cell_box<PW>(my_ok, my_ok ? my_nx : 0.f, my_ok ? my_ny : 0.f, xlo, ylo, W, H);
    bool tab_bad = false;
    constexpr int NTAB = (2 * D + kWaves - 1) / kWaves;
#pragma unroll
    for (int n = 0; n < NTAB; ++n) {
        const int a = wave + n * kWaves;   // scalar
        if (a < 2 * D) {
            const bool isy = a >= D;
            const int k = isy ? a - D : a;
            const float lin = isy ? gfn::linspace_step_at(ylo, yhi, p.win_ystep, D, k) : gfn::linspace_step_at(xlo, xhi, p.win_xstep, D, k);
            const float pix = unnorm((isy ? my_ny : my_nx) + lin, isy ? H : W);
            const float fl = floorf(pix);
            const int origin = isy ? c.Y0 : c.X0;
            tab_bad |= (origin != kFar) & !(fl == (float)(origin + k));
            tab[lane * TS + a] = pix - fl;
        }
    }
    if (wave == 0) {
        cellX0[lane] = c.X0;
        cellY0[lane] = c.Y0;
        cellNx[lane] = my_ok ? my_nx : 0.f;
        cellNy[lane] = my_ok ? my_ny : 0.f;
        cellFlag[lane] = c.flag;
        const unsigned long long slow_mask = __ballot(c.flag == kCellSlow);
        if (lane == 0) hdr[4] = __popcll(slow_mask);
    }
    {
        // channels NF0 wave .. NF0 wave + NF0 - 1 of cell fcell (one chunk: NF0 divides 16): NF0 bf16 into the hi half of the
        // cell's slot of that chunk, NF0 into the lo half -- one store each
        const int fcell = ((fc >> 3) << 5) | (fr << 3) | (fc & 7);
        const int ch0 = wave * NF0;  // scalar
        unsigned hi[NF0 / 2], lo[NF0 / 2];
#pragma unroll
        for (int k = 0; k < NF0 / 2; ++k) split_pair(fok ? f0v[2 * k] : 0.f, fok ? f0v[2 * k + 1] : 0.f, hi[k], lo[k]);
        unsigned *slot = reinterpret_cast<unsigned *>(f0b + fcell * kF0Cell + (ch0 >> 4) * 64 + (ch0 & 15) * 2);
        if constexpr (NF0 == 2) {
            slot[0] = hi[0]; slot[8] = lo[0];
        } else if constexpr (NF0 == 4) {
            *reinterpret_cast<u32x2_t *>(slot) = u32x2_t{hi[0], hi[1]};
            *reinterpret_cast<u32x2_t *>(slot + 8) = u32x2_t{lo[0], lo[1]};
        } else {
            static_assert(NF0 == 8, "C is 16, 32 or 64");
            *reinterpret_cast<i32x4 *>(slot) = make_i32x4((int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]);
            *reinterpret_cast<i32x4 *>(slot + 8) = make_i32x4((int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3]);
        }
    }
    // this wave's group: box of its 16 cells' windows, relative to the region they are staged in
    const int g = wave >> 1, sub = wave & 1;                       // scalars
    const RowPlan &ug = (HALVES && g >= 2) ? uB : uA;
    int gx0, gy0, nb;
    {
        const int rx0 = row_min_i32(c.bx0), ry0 = row_min_i32(c.by0), ry1 = row_min_i32(-c.by1);
        const int l15 = g * 16 + 15;
        const int bx0 = __builtin_amdgcn_readlane(rx0, l15), by0 = __builtin_amdgcn_readlane(ry0, l15), by1 = -__builtin_amdgcn_readlane(ry1, l15);
        const bool any = bx0 != kFar;
        gx0 = any ? bx0 - ug.x0 + 16 * sub : 0;
        gy0 = any ? by0 - ug.y0 : 0;
        nb = any ? min(by1 - by0, NBW) : 0;  // the plan guarantees <= NBW
    }
    STAMP(2);
    quad_commit_mm<PRE, CHECK, FT>(smem, pre, H, W, uA, wave, lane, qlA, 0);
    quad_rest_mm<CHECK, FT>(smem, f1r, 0u, H, W, uA, wave, lane, qlA, PRE);
    STAMP(3);
    __syncthreads();
    STAMP(4);
    if (tab_bad && atomicOr(&cellFlag[lane], kCellSlow) == 0) atomicAdd(&hdr[4], 1);  // rare

    // ---- matrix-core D-stage ------------------------------------------------------------------------------------------------
    // lane = position m (lane & 15) of the block + 16 * k-group q: the A operand is bytes 16 q .. 16 q + 15 of slot (row, gx0 + m)
    const int mq = lane >> 4, mm = lane & 15;
    const unsigned a_addr = (unsigned)((gy0 * ug.pitch + gx0 + mm) * (kSlotV4 * 16) + mq * 16);
    const unsigned a_step = (unsigned)(ug.pitch * (kSlotV4 * 16));
    const unsigned b_addr = (unsigned)((g * 16 + mm) * kF0Cell + (mq & 1) * 16);
    f32x4 acc[NBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    STAMP(5);

    constexpr int NS = HALVES ? 2 * NCH : NCH;  // steps: (half, chunk)
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        const int ch = HALVES ? st % NCH : st;
        const int half = HALVES ? st / NCH : 0;
        const bool more = st + 1 < NS;
        const int nch = HALVES ? (st + 1) % NCH : st + 1, nhalf = HALVES ? (st + 1) / NCH : 0;
        const unsigned next_off = (unsigned)(nch * kChunk) * (unsigned)(H * W) * (unsigned)sizeof(FT);
        const RowPlan &un = (HALVES && nhalf == 1) ? uB : uA;
        const QuadLane &qn = (HALVES && nhalf == 1) ? qlB : qlA;
        if (more) quad_issue<PRE, CHECK, FT, kSlot8>(pre, f1r, next_off, H, W, un, wave, lane, qn, 0);  // in flight across the products
        if (!HALVES || (g >> 1) == half) {  // scalar
            const bf16x8_t b1 = *reinterpret_cast<const bf16x8_t *>(f0b + b_addr + ch * 64);
            const bf16x8_t b2 = *reinterpret_cast<const bf16x8_t *>(f0b + b_addr + ch * 64 + 32);
            // four blocks at a time: their A operands are requested together, then the eight products issue back to back (a
            // branch per block made every block a serial LDS round trip + two dependent instructions); rows past the group's
            // box repeat its last row into accumulators nobody reads
#pragma unroll
            for (int i0 = 0; i0 < NBW; i0 += 4) {
                if (i0 < nb) {  // scalar
                    bf16x8_t a[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        a[j] = *reinterpret_cast<const bf16x8_t *>(smem + a_addr + (unsigned)min(i0 + j, nb - 1) * a_step);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], b1, acc[i0 + j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], b2, acc[i0 + j], 0, 0, 0);
                }
            }
        }
        STAMP(st == 0 ? 6 : 9);
        if (more) {
            __syncthreads();  // everyone is done reading this step's pixels
            STAMP(7);
            quad_commit_mm<PRE, CHECK, FT>(smem, pre, H, W, un, wave, lane, qn, 0);
            quad_rest_mm<CHECK, FT>(smem, f1r, next_off, H, W, un, wave, lane, qn, PRE);
            __syncthreads();
            STAMP(8);
        }
    }

    // ---- accumulators -> D buffer (aliases the stage) ---------------------------------------------------------------------------
    __syncthreads();
    STAMP(10);
    {
        // the lane's cell and its four positions of block i: region row gy0 + i, columns gx0 + 4 q .. + 3
        const int cell = g * 16 + mm;
        const int X0 = cellX0[cell], Y0 = cellY0[cell];
        const bool has = X0 != kFar;                                  // false: off the grid, flagged, or its window misses the image
        const int dx0 = has ? gx0 + 4 * mq - (X0 - ug.x0) : -1000;   // window column of the first of the four
        const int dy0 = has ? gy0 - (Y0 - ug.y0) : 0;                // window row of block 0
        const bool col_ok = (unsigned)(dx0 + 3) < (unsigned)(PW + 3);
        float *d = dbuf + cell * DS + 3 + dy0 * RP + dx0;
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
            if ((i & ~3) < nb) {  // scalar, per group of four blocks as above (rows past the box fail every cell's row test)
                if (col_ok & ((unsigned)(dy0 + i) < (unsigned)PW)) {
                    d[i * RP + 0] = acc[i][0];
                    d[i * RP + 1] = acc[i][1];
                    d[i * RP + 2] = acc[i][2];
                    d[i * RP + 3] = acc[i][3];
                }
            }
        }
    }
    STAMP(11);
    __syncthreads();
    STAMP(12);
    {
        const int er = lane >> 4, ec = lane & 15;
        const int cell = ((ec >> 3) << 5) | (er << 3) | (ec & 7);
        const int gi = row0 + er, gj = col0 + ec;
        const int flag = cellFlag[cell];
        if ((gi < G) & (gj < G) & !(flag & kCellSlow)) {
            const bool empty = (flag & kCellEmpty) != 0;
            const float *dc = dbuf + cell * DS + 3;
            const float *tc = tab + cell * TS;
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
                    const float wy1 = tc[D + ky];
                    const float wy1s = wy1 * p.inv_sqrt_c, wy0s = (1.f - wy1) * p.inv_sqrt_c;
                    const float *dd = dc + ky * RP;
                    float m[PW];
#pragma unroll
                    for (int x = 0; x < PW; ++x) m[x] = fmaf(dd[RP + x], wy1s, dd[x] * wy0s);
#pragma unroll
                    for (int kx = 0; kx < D; ++kx) {
                        const float val = fmaf(m[kx + 1], wx1[kx], m[kx] * wx0[kx]);
                        buf_st_nt(outr, goff, (unsigned)(ky * D + kx) * GG4, empty ? 0.f : val);
                    }
                }
            }
        }
    }
    STAMP(13);
#ifdef GFN_ABLATE
    if (stamping)
        printf("mm r%d wave %d (cycles from entry): all issued %lld | cells+table+f0 in LDS %lld | stage0 committed %lld | barrier %lld | set-up %lld | "
               "D0 %lld | barrier %lld | stage1 committed+barrier %lld | D1 %lld | barrier %lld | dbuf %lld | barrier %lld | stores issued %lld\n",
               R, tid >> 6, stamp[1] - stamp[0], stamp[2] - stamp[0], stamp[3] - stamp[0], stamp[4] - stamp[0], stamp[5] - stamp[0], stamp[6] - stamp[0],
               stamp[7] - stamp[0], stamp[8] - stamp[0], stamp[9] - stamp[0], stamp[10] - stamp[0], stamp[11] - stamp[0], stamp[12] - stamp[0],
               stamp[13] - stamp[0]);
#endif

    // ---- flagged cells: general per-tap routine (about one cell in 10^4) ---------------------------------------------
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

template <int R, int NCH, typename FT>
__global__ __launch_bounds__(kThreads, Lean<R>::kMinWaves) void local_corr_tile_mm_kernel(LcParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kLeanWorkers = lean_workers<R>();
    if constexpr (kLeanWorkers > 0) {
        if (blockIdx.x < kLeanWorkers) {  // block-uniform
            second_launch_worker<R, 2, FT, Lean<R>::kStage>(p, smem, (int)blockIdx.x, kLeanWorkers);
            return;
        }
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned wid = gfn::xcd_remap(blockIdx.x - kLeanWorkers, gridDim.x - kLeanWorkers);
    typedef int i32x8 __attribute__((ext_vector_type(8)));
    i32x8 pl;
    {
        const int *pp = p.plan + (size_t)wid * kPlanInts;
        asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pl) : "s"(pp) : "memory");
    }
    const int flags = pl[3];
    if (flags & kPlanSecond) return;
    RowPlan uA, uB;
    uA.x0 = pl[0]; uA.y0 = pl[1]; uA.w = pl[2] & 0xffff; uA.h = pl[2] >> 16;
    uB.x0 = pl[4]; uB.y0 = pl[5]; uB.w = pl[6] & 0xffff; uB.h = pl[6] >> 16;
    (void)region_fits<R>(uA);
    (void)region_fits<R>(uB);
    const bool interior = (flags & kPlanInterior) != 0;
    if (flags & kPlanHalves) {
        if (interior) lean_tile_mm<R, NCH, false, true, FT>(p, smem, uA, uB, wid, tid, lane, wave);
        else lean_tile_mm<R, NCH, true, true, FT>(p, smem, uA, uB, wid, tid, lane, wave);
    } else {
        if (interior) lean_tile_mm<R, NCH, false, false, FT>(p, smem, uA, uB, wid, tid, lane, wave);
        else lean_tile_mm<R, NCH, true, false, FT>(p, smem, uA, uB, wid, tid, lane, wave);
    }
}
