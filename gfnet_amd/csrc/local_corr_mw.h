// local_corr_mw.h -- round 3: the r = 3 / 4 local-correlation tile on the matrix core, third shape: FOUR waves per workgroup, a wave per
// group.  Included by local_corr.hip after local_corr_mq.h; built only where GFN_MM_DEFAULT == 1 (libgfnet_hip_mm.so).
//
// What the two earlier shapes ran into (profiles/r03_local_corr_mm.md): a 4 x 16-cell tile has four groups of 2 x 8 cells whose boxes
// are ~13 rows x 2 column tiles = 26 blocks = 104 accumulator registers per group.  With eight waves in the workgroup and two
// workgroups per CU (4 waves per SIMD: 128 registers each) the accumulators spill; with sixteen waves in ONE workgroup per CU they fit
// but every phase runs in lockstep across the CU.  This shape keeps two workgroups per CU and halves the waves instead: 2 x 4 waves per
// CU = 2 waves per SIMD = 256 registers per wave.  A wave owns a group: <= 32 blocks = 128 accumulator registers, alive across the
// channel chunks; every staging load of the tile (all chunks: 96 registers at C = 32) is in flight before anything else happens,
// because the accumulators only come alive after the first chunk is filed.
//   plan: the matrix-core plan of local_corr_lean.h (one unclipped region per tile, groups <= 32 columns x kMwRows rows, else the list)
//   LDS : stage (64-byte swizzled slots of 16 channels as hi | lo octets) aliased by the guarded D buffer, cells, fraction table, f0
//         block as bf16 hi / lo: 80 KB
// The staging helpers, swizzle, operand layout and epilogue arithmetic are local_corr_mm.h's.
//
// Outcome (profiles/r03_local_corr_mm.md): no spills (223 VGPRs), parity green, 104.4 / 154.9 us for the two r = 4 calls of the probe
// against the lean fp32 kernel's 92.5 / 138.0 (the persistent kernel: 123) -- PARKED like its predecessors, the kernel of
// -DGFN_MM_DEFAULT=1 builds (-DGFN_MM_DEFAULT=2: the persistent kernel).  A tile takes ~23 k cycles (lean: 22 k): 4 k to issue, chunk 0
// filed at 8-9 k, products 2 x 2.6 k (52 instructions of 16 cycles each: a third of that), the second chunk's split 2 k, filing the
// accumulators 3.3 k, stores 2.5 k.  Every phase runs 3-4 x its issue-limited time: with two waves per SIMD nothing hides the LDS and
// dependent-instruction latencies inside a wave, and per tile the four waves execute as many vector instructions as the lean kernel's
// eight (the FMAs went, the bf16 split and the filing of 104 accumulator quads per wave came).  What was tried on the way: operand
// reads one row ahead in ping-pong registers (products 3.3 -> 2.6 k), blocks outside a cell's window written to a dump word instead of
// a divergent branch per block, no masks at the commit of interior tiles, the fraction table before the first barrier (level: the
// loads have arrived by then).

constexpr int kMwThreads = 256, kMwWaves = 4;
constexpr int kMwPre = 3;   // items of a chunk a wave holds in registers (768 positions; more go through mm_rest)

__device__ __forceinline__ int mw_dpos(int cell) { return ((cell & 31) << 1) | (cell >> 5); }

// between two chunks' products: everyone is done reading chunk ch's pixels, chunk ch + 1 (in registers since the top of the tile) is filed
template <int NCH, typename F>
__device__ __forceinline__ void mw_next_chunk(int ch, F &commit) {
    if constexpr (NCH > 1) {
        if (ch == 0) { __syncthreads(); commit(std::integral_constant<int, 1>{}); __syncthreads(); }
    }
    if constexpr (NCH > 2) {
        if (ch == 1) { __syncthreads(); commit(std::integral_constant<int, 2>{}); __syncthreads(); }
        if (ch == 2) { __syncthreads(); commit(std::integral_constant<int, 3>{}); __syncthreads(); }
    }
}

template <int R, int C, typename FT>
__global__ __launch_bounds__(kMwThreads, 2) void local_corr_mw_kernel(LcParams p) {
    typedef Mw<R, C> M;
    static_assert(M::kDbufBytes <= M::kStage, "the D buffer aliases the stage");
    constexpr int PW = M::PW, D = M::D, K = M::K, TS = M::TS, NC = M::NC, RP = M::RP, DS = M::DS, NBW = M::NBW, NCH = M::NCH, NW = M::NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *dbuf = reinterpret_cast<float *>(smem);
    unsigned char *misc = smem + M::kStage;
    int *cellX0 = reinterpret_cast<int *>(misc);
    int *cellY0 = cellX0 + NC;
    float *cellNx = reinterpret_cast<float *>(cellY0 + NC);
    float *cellNy = cellNx + NC;
    int *cellFlag = reinterpret_cast<int *>(cellNy + NC);
    int *hdr = cellFlag + NC;   // [4] flagged cells; [8 + 4 g ..] group box x0, y0, y1
    float *tab = reinterpret_cast<float *>(misc + M::kCellBytes);
    unsigned char *f0b = misc + M::kCellBytes + M::kTabBytes;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tid = threadIdx.x;
    const int G = p.G, H = p.H, W = p.W;
    const float xhi = p.win_xhi, xlo = -xhi, yhi = p.win_yhi, ylo = -yhi;
    const unsigned GG4 = (unsigned)(G * G) * 4u;
    const unsigned wid = gfn::xcd_remap(blockIdx.x, gridDim.x);
    // the plan came through a vector load of a uniform address: say so (scalar offsets of the buffer loads, uniform branches)
    const int4 pl = reinterpret_cast<const int4 *>(p.plan)[kPlanV4 * wid];
    const int pflags = __builtin_amdgcn_readfirstlane(pl.w);
    if (pflags & kPlanSecond) return;  // on the second launch's list
    MmRegion u;
    u.x0 = __builtin_amdgcn_readfirstlane(pl.x); u.y0 = __builtin_amdgcn_readfirstlane(pl.y);
    {
        const int hw = __builtin_amdgcn_readfirstlane(pl.z);
        u.w = hw & 0xffff; u.h = hw >> 16;
    }
    mm_region_geometry(u);
    const int tiles = p.tiles_x * p.tiles_y;
    const int b = wid / tiles, tile = wid - b * tiles;
    const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    const int row0 = ty * 4, col0 = tx * kTileW;
#ifdef GFN_ABLATE
    const bool stamping = ABL(p, 512) && (blockIdx.x % 1999) == 1000 && (tid & 63) == 0 && (tid >> 6) < 2;
    long long stamp[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    STAMP(0);

    // ---- every load of the tile goes out now -------------------------------------------------------------------------------------------
    const int nitems = (u.h * u.nq + 15) >> 4;          // items (16 quads x 16 channels) of a chunk
    const int ipw = (nitems + NW - 1) / NW;              // ... per wave
    const rsrc_t f1r = make_rsrc(f1_of<FT>(p, b), (unsigned)C * (unsigned)(H * W) * (unsigned)sizeof(FT));
    const unsigned chunk_off = 16u * (unsigned)(H * W) * (unsigned)sizeof(FT);
    const unsigned plane4 = (unsigned)(H * W) * (unsigned)sizeof(FT);
    MmItem it[kMwPre];
#pragma unroll
    for (int n = 0; n < kMwPre; ++n) it[n] = mm_item<M, NW, true, FT, true>(u, H, W, wave, lane, n);
    typename QuadRaw<FT>::type pre[NCH][kMwPre][4];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int n = 0; n < kMwPre; ++n) {
            const unsigned vo = n < ipw ? it[n].voff : kOffRange;
#pragma unroll
            for (int j = 0; j < 4; ++j) pre[ch][n][j] = QuadRaw<FT>::load(f1r, vo, (unsigned)ch * chunk_off + (unsigned)j * plane4);
        }
    // the f0 block: lane = channel (lane & 3) x quad of cells (tile row (lane >> 4), four columns 4 ((lane >> 2) & 3) ..); wave w
    // takes channels (C / 4) w .. in NF0 loads of four channels each
    constexpr int NF0 = C / 16;
    const int fk = lane & 3, fr = lane >> 4, fc4 = ((lane >> 2) & 3) * 4;
    f32x4 f0v[NF0];
    {
        const bool in = (row0 + fr < G) & (col0 + fc4 < G);
        const rsrc_t f0r = make_rsrc(p.f0 + (size_t)b * p.f0_bs, (unsigned)C * GG4);
#pragma unroll
        for (int l = 0; l < NF0; ++l) {
            const unsigned off = in ? (unsigned)((row0 + fr) * G + col0 + fc4) * 4u + (unsigned)(wave * (C / 4) + l * 4 + fk) * GG4 : kOffRange;
            f0v[l] = buf_ld4(f0r, off, 0u);
        }
    }
    float my_nx = 0.f, my_ny = 0.f;
    const int my_gi = row0 + cell_row(lane), my_gj = col0 + cell_col(lane);
    const bool my_ok = (my_gi < G) & (my_gj < G);
    {   // lane = cell id; every wave (6 more loads per tile): its share of the fraction table is worked out under the stage loads
        const rsrc_t flr = make_rsrc(p.flow + (size_t)b * 2 * G * G, 2u * GG4);
        const unsigned fo = my_ok ? (unsigned)(my_gi * G + my_gj) * 4u : kOffRange;
        my_nx = buf_ld(flr, fo, 0u);
        my_ny = buf_ld(flr, fo, GG4);
    }
    STAMP(1);

    // ---- cells and group boxes (wave 0), fraction table, f0 block -> bf16 hi / lo -----------------------------------------------------------
    const CellBox c = cell_box<PW>(my_ok, my_ok ? my_nx : 0.f, my_ok ? my_ny : 0.f, xlo, ylo, W, H);
    if (wave == 0) {  // scalar
        cellX0[lane] = c.X0;
        cellY0[lane] = c.Y0;
        cellNx[lane] = my_ok ? my_nx : 0.f;
        cellNy[lane] = my_ok ? my_ny : 0.f;
        cellFlag[lane] = c.flag;
        const unsigned long long slow_mask = __ballot(c.flag == kCellSlow);
        if (lane == 0) hdr[4] = __popcll(slow_mask);
        const int rx0 = row_min_i32(c.bx0), ry0 = row_min_i32(c.by0), ry1 = row_min_i32(-c.by1);
        if ((lane & 15) == 15) {
            int *gb = hdr + 8 + (lane >> 4) * 4;
            gb[0] = rx0; gb[1] = ry0; gb[2] = -ry1;
        }
    }
    {
        // lanes fk and fk ^ 1 exchange (DPP quad_perm 1, 0, 3, 2), so that each files channel PAIRS (32-bit writes) of two of the four cells
        const int odd = fk & 1;
#pragma unroll
        for (int l = 0; l < NF0; ++l) {
            float other[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float own = f0v[l][e];   // (bit_cast on a vector element reads element 0 under hipcc 7.2: through a scalar)
                other[e] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, own), 0xB1, 0xf, 0xf, false));
            }
            const int ch = wave * (C / 4) + l * 4 + (fk & ~1);   // even channel of the pair
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
                const int fc = fc4 + 2 * odd + e2;
                const bool fok = (row0 + fr < G) & (col0 + fc < G);
                const float mine = odd ? (e2 ? f0v[l][3] : f0v[l][2]) : (e2 ? f0v[l][1] : f0v[l][0]);
                const float theirs = odd ? (e2 ? other[3] : other[2]) : (e2 ? other[1] : other[0]);
                unsigned hi, lo;
                split_pair(fok ? (odd ? theirs : mine) : 0.f, fok ? (odd ? mine : theirs) : 0.f, hi, lo);
                const int fcell = ((fc >> 3) << 5) | (fr << 3) | (fc & 7);
                unsigned *slot = reinterpret_cast<unsigned *>(f0b + fcell * M::kF0Cell + (ch >> 4) * 64 + (ch & 15) * 2);
                slot[0] = hi;
                slot[8] = lo;
            }
        }
    }
    // fraction table: lane = cell, wave = tap index (read by the epilogue, barriers from here)
    bool tab_bad = false;
    {
        const float cnx = my_ok ? my_nx : 0.f, cny = my_ok ? my_ny : 0.f;
        const int cX0 = c.X0, cY0 = c.Y0;
        constexpr int NTAB = (2 * D + NW - 1) / NW;
#pragma unroll
        for (int n = 0; n < NTAB; ++n) {
            const int a = wave + n * NW;   // scalar
            if (a < 2 * D) {
                const bool isy = a >= D;
                const int k = isy ? a - D : a;
                const float lin = isy ? gfn::linspace_step_at(ylo, yhi, p.win_ystep, D, k) : gfn::linspace_step_at(xlo, xhi, p.win_xstep, D, k);
                const float pix = unnorm((isy ? cny : cnx) + lin, isy ? H : W);
                const float fl = floorf(pix);
                const int origin = isy ? cY0 : cX0;
                tab_bad |= (origin != kFar) & !(fl == (float)(origin + k));
                tab[lane * TS + a] = pix - fl;
            }
        }
    }
    // chunk 0 -> LDS
    const bool interior = (pflags & kPlanInterior) != 0;   // scalar: no staged pixel lies outside the image -> no masks at the commit
    auto commit = [&](auto chc) {   // (the chunk as a type: pre[] must only ever be indexed by constants)
        constexpr int ch = decltype(chc)::value;
#pragma unroll
        for (int n = 0; n < kMwPre; ++n) {
            const unsigned meta = it[n].meta;
            if ((n < ipw) & ((meta >> 22) & 1u)) {
                if (interior) mm_commit_one<M, false, FT>(smem, pre[ch][n], meta);
                else mm_commit_one<M, true, FT>(smem, pre[ch][n], meta);
            }
        }
        // regions of more than 768 positions: the remaining items, loaded here
        for (int k = kMwPre; k < ipw; ++k) {
            const MmItem ix = mm_item<M, NW, true, FT>(u, H, W, wave, lane, k);
            typename QuadRaw<FT>::type a[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = QuadRaw<FT>::load(f1r, ix.voff, (unsigned)ch * chunk_off + (unsigned)j * plane4);
            if ((ix.meta >> 22) & 1u) mm_commit_one<M, true, FT>(smem, a, ix.meta);
        }
    };
    commit(std::integral_constant<int, 0>{});
    STAMP(2);
    __syncthreads();
    STAMP(3);
    if (tab_bad && atomicOr(&cellFlag[lane], kCellSlow) == 0) atomicAdd(&hdr[4], 1);  // rare
    STAMP(8);
    // ---- products: this wave's group, block i = (row i >> 1 of the group's box, column tile i & 1) -----------------------------------------
    const int g = wave;
    int gx0, gy0, nb;
    {
        const int bx0 = __builtin_amdgcn_readfirstlane(hdr[8 + 4 * g]), by0 = __builtin_amdgcn_readfirstlane(hdr[9 + 4 * g]),
                  by1 = __builtin_amdgcn_readfirstlane(hdr[10 + 4 * g]);
        const bool any = bx0 != kFar;
        gx0 = any ? bx0 - u.x0 : 0;
        gy0 = any ? by0 - u.y0 : 0;
        nb = any ? min(2 * (by1 - by0), NBW) : 0;   // the plan guarantees <= NBW
    }
    const int mq = lane >> 4, mm = lane & 15;
    const unsigned slot0 = (unsigned)(gy0 * u.pitch + gx0 + mm);
    const unsigned b_addr = (unsigned)((g * 16 + mm) * M::kF0Cell + (mq & 1) * 16);
    f32x4 acc[NBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const bf16x8_t b1 = *reinterpret_cast<const bf16x8_t *>(f0b + b_addr + ch * 64);
        const bf16x8_t b2 = *reinterpret_cast<const bf16x8_t *>(f0b + b_addr + ch * 64 + 32);
        unsigned slot_c = slot0;
        asm volatile("" : "+v"(slot_c));   // operand addresses re-derived per chunk
        auto a_op = [&](int i) {
            const unsigned s = slot_c + (unsigned)(i >> 1) * (unsigned)u.pitch + (unsigned)(i & 1) * 16u;
            return *reinterpret_cast<const bf16x8_t *>(smem + s * (unsigned)M::SLOT + (((unsigned)mq ^ mm_swz<16>(s)) << 4));
        };
        // a row's two operands are read while the previous row's four products run (two waves per SIMD hide no LDS latency): two
        // register sets in turn, no copies (a copy made the compiler wait for the read right behind the products); the row past the
        // box repeats the last one
        const int last = max(nb - 2, 0);
        bf16x8_t aA0 = a_op(0), aA1 = a_op(1), aB0, aB1;
#pragma unroll
        for (int i0 = 0; i0 < NBW; i0 += 4) {
            if (i0 < nb) {  // scalar (nb is even: both column tiles of a row)
                const int nx = min(i0 + 2, last);
                aB0 = a_op(nx); aB1 = a_op(nx + 1);
                acc[i0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aA0, b1, acc[i0], 0, 0, 0);
                acc[i0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aA1, b1, acc[i0 + 1], 0, 0, 0);
                acc[i0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aA0, b2, acc[i0], 0, 0, 0);
                acc[i0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aA1, b2, acc[i0 + 1], 0, 0, 0);
            }
            if (i0 + 2 < nb) {
                const int nx = min(i0 + 4, last);
                aA0 = a_op(nx); aA1 = a_op(nx + 1);
                acc[i0 + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aB0, b1, acc[i0 + 2], 0, 0, 0);
                acc[i0 + 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aB1, b1, acc[i0 + 3], 0, 0, 0);
                acc[i0 + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aB0, b2, acc[i0 + 2], 0, 0, 0);
                acc[i0 + 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aB1, b2, acc[i0 + 3], 0, 0, 0);
            }
        }
        STAMP(9 + 2 * ch);
        mw_next_chunk<NCH>(ch, commit);
        STAMP(10 + 2 * ch);
    }
    STAMP(4);
    __syncthreads();  // the D buffer aliases the stage
    {
        const int cellg = g * 16 + mm;
        const int X0 = cellX0[cellg], Y0 = cellY0[cellg];
        const bool has = X0 != kFar;                                  // false: off the grid, flagged, or its window misses the image
        const int dxa = has ? gx0 + 4 * mq - (X0 - u.x0) : -1000;    // window column of the first of the lane's four positions, column tile 0
        const int dy0 = has ? gy0 - (Y0 - u.y0) : 0;                 // window row of the box's first row
        const bool ok0 = (unsigned)(dxa + 3) < (unsigned)(PW + 3), ok1 = (unsigned)(dxa + 16 + 3) < (unsigned)(PW + 3);
        float *dwin = dbuf + mw_dpos(cellg) * DS + 3 + dy0 * RP + dxa;
        float *dump = reinterpret_cast<float *>(hdr + 24);   // four floats nobody reads: a block outside the cell's window lands here (no
                                                             // divergent branch per block)
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
            if (i < nb) {  // scalar
                const int row = i >> 1, ct = i & 1;
                const bool in = (ct ? ok1 : ok0) & ((unsigned)(dy0 + row) < (unsigned)PW);
                float *q = in ? dwin + row * RP + ct * 16 : dump;
                q[0] = acc[i][0];
                q[1] = acc[i][1];
                q[2] = acc[i][2];
                q[3] = acc[i][3];
            }
        }
    }
    STAMP(5);
    __syncthreads();
    STAMP(6);
    {
        // lane -> cell so that a wave stores whole 64-byte grid-row segments; wave = tap row ky
        const int er = lane >> 4, ec = lane & 15;
        const int cell = ((ec >> 3) << 5) | (er << 3) | (ec & 7);
        const int gi = row0 + er, gj = col0 + ec;
        const int flag = cellFlag[cell];
        if ((gi < G) & (gj < G) & !(flag & kCellSlow)) {
            const bool empty = (flag & kCellEmpty) != 0;
            const float *dc = dbuf + mw_dpos(cell) * DS + 3;
            const float *tc = tab + cell * TS;
            const unsigned goff = (unsigned)(gi * G + gj) * 4u;
            const rsrc_t outr = make_rsrc(p.out + (size_t)b * p.out_bs, (unsigned)K * GG4);
            constexpr int NR = (D + NW - 1) / NW;
#pragma unroll
            for (int n = 0; n < NR; ++n) {
                const int ky = wave + n * NW;  // scalar
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
                        buf_st_nt(outr, goff, (unsigned)(ky * D + kx) * GG4, empty ? 0.f : val);
                    }
                }
            }
        }
    }
    STAMP(7);
    // ---- flagged cells: general per-tap routine (about one cell in 10^4) -----------------------------------------------------------
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
        const int totalk = hdr[4] * K;
        for (int e = tid; e < totalk; e += kMwThreads) {
            const int cell = cellX0[e / K], k = e % K;
            const int gi = row0 + cell_row(cell), gj = col0 + cell_col(cell);
            p.out[(size_t)b * p.out_bs + ((size_t)k * G + gi) * G + gj] =
                tap_general<FT>(p, b, gi, gj, k / D, k % D, D, cellNx[cell], cellNy[cell]);
        }
    }
#ifdef GFN_ABLATE
    if (stamping)
        printf("mw r%d wave %d (cycles): all issued %lld | chunk 0 filed %lld | barrier %lld | table %lld | products 0 %lld | chunk 1 filed %lld | "
               "products 1 %lld | filed %lld | barrier %lld | stores issued %lld\n",
               R, tid >> 6, stamp[1] - stamp[0], stamp[2] - stamp[0], stamp[3] - stamp[0], stamp[8] - stamp[0], stamp[9] - stamp[0],
               stamp[10] - stamp[0], stamp[11] - stamp[0], stamp[5] - stamp[0], stamp[6] - stamp[0], stamp[7] - stamp[0]);
#endif
}
