// local_corr_mm.h -- round 3: the local-correlation tile with its D-stage on the matrix core.  Included by local_corr.hip after
// local_corr_lean.h (same namespace: cell boxes, the fraction table, buffer addressing and the epilogue arithmetic are the lean
// path's).
//
// Why (profiles/local_corr_sq_pmc.json, round 2): at r = 4 a wave of the lean kernel issued 949 vector instructions of which 448
// were the D-stage's v_fma_f32, every one of them fed by its own LDS dword: FMA issue and the D-stage's LDS reads were each worth
// a third of the kernel and ran one after the other behind barriers.  v_pk_fma_f32 issues at half rate on gfx950 (tools/micro/
// valu_rate.hip), so the products have to leave the VALU.
//
// The product: D[cell][position] = sum_c f0[cell][c] * f1[c][position] is (positions x channels) . (channels x cells).  Per
// group of 2 x 8 cells (16 cells = the N of v_mfma_f32_16x16x32_bf16, one DPP row of the tile's 64 cell lanes) the positions its
// windows touch are a box of <= 32 columns x <= 2 NBW rows of the staged region; a block = 16 consecutive positions of one
// region row (M) x the 16 cells.  fp32 accuracy from bf16 operands: every value is split x = hi + lo when it is filed in LDS
// (both pieces round-to-nearest bf16: residual <= 2^-18 |x|), and the K = 32 of the instruction holds 16 channels as [hi | lo]:
//      A (positions) = [f1_hi(16) | f1_lo(16)],   B1 (cells) = [f0_hi | f0_hi],   B2 = [f0_lo | f0_lo]
//      mfma(A, B1) + mfma(A, B2) = sum_c (f1_hi + f1_lo)(f0_hi + f0_lo):  all four partial products, fp32 accumulation,
// a product off by <= 2^-17 relative (tests: 1e-4 * max(1, |ref|) against the oracle; measured a few 1e-6).  fp16 maps split
// exactly.  About a quarter of the products fall inside some cell's window: the matrix core has 16x the VALU's rate, and LDS feeds
// it one dword per 64 products instead of one per product.
//
// The first cut kept the lean kernel's shape (two 8-wave workgroups per CU, 16-channel chunks, accumulators kept across the
// chunks): 64 accumulator registers per wave beside the 32 of the next chunk's loads in flight spilled, and a 64-way bank
// conflict in the f0 filing hid the rest (118 us against the lean kernel's 99).
// This version gives the CU to ONE 16-wave workgroup:
//   * all channels of a position are staged at once (up to 32: 128-byte slots; 64-channel maps take two passes), so a block is
//     complete after 2 x KC/16 instructions and is filed in the D buffer at once: 8-16 accumulator registers alive per wave (two
//     passes: 4 NBW = 40, kept across them) -- four waves serve a group (column tile x row parity);
//   * slots carry no padding: the eight 16-byte pieces of a slot (hi/lo x channel octet) are XOR-swizzled with bits of the slot
//     index, which makes both the A-operand read (16 consecutive slots, one piece pair) and the staging writes conflict-free;
//     ~800 positions fit beside the D buffer (single-pass shapes), so the "halves" mode is gone;
//   * the D buffer (rows of PW + 6 floats: three guard columns either side, so a lane's four consecutive positions need one range
//     test) has LDS of its own (~100 KB stage + 41 KB D buffer at r = 4): accumulators leave right behind their products;
//   * the workgroup is PERSISTENT: it walks tiles bid, bid + gridDim, ... of an XCD-contiguous order, and the next tile's loads
//     (flow, f0 block, f1 quads: ~40 registers) are in flight under the current tile's products and epilogue -- with one
//     workgroup per CU nothing else would cover the L2 round trip and the ~12 B/clk at which a CU's staging loads drain
//     (measured one tile per workgroup: 13 k cycles per tile of which 7 k waiting for the loads).
//
// Outcome (profiles/r03_local_corr_mm.md): every phase is short (products 1.8-2.5 k cycles against the fp32 D-stage's 2 x 2.8-4.5 k),
// but one workgroup per CU runs them in lockstep and the issue of a tile's ~190 staging loads alone takes 5-6 k cycles: 123 us
// for the scale-4 op against the lean fp32 kernel's 91-98.  PARKED: built only with -DGFN_MM_DEFAULT=1 (libgfnet_hip_mm.so, r = 4
// on 32-channel and r = 3 on 16-channel maps), kept parity-green by tests/test_local_corr_mm_gpu.py.
//
// Numerics class: NOT bit-identical to the fp32 FMA kernels (variant 4: the round-2 lean kernel, variant 2: round 1), which stay
// as cross-checks.

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

constexpr int kMmThreads = 1024, kMmWaves = 16;
constexpr int kMmLds = 160 * 1024;
// kMmNBW (local_corr_lean.h) = blocks (accumulator quads) per wave: a group's box is at most 2 * kMmNBW rows high

template <int R, int C>
struct Mm {
    static constexpr int KC = C < 32 ? C : 32;   // channels staged per pass
    static constexpr int NPASS = C / KC;
    static constexpr int NSUB = KC / 16;         // 16-channel sub-chunks of a pass: one [hi | lo] K = 32 each
    static constexpr int NPIECE = KC / 4;        // 16-byte pieces per slot: hi octets, then lo octets
    static constexpr int SLOT = KC * 4;          // bytes per staged position
    static constexpr int PW = 2 * R + 2, D = 2 * R + 1, K = D * D, TS = 2 * D + 1;
    static constexpr int NC = 64;
    static constexpr int RP = PW + 6;                           // D-buffer row: 3 guard floats | PW positions | 3 guard floats
    static constexpr int DS = ((PW * RP + 31) & ~31) + 5;       // == 5 (mod 32): neighbouring cells' windows start ~1.75 floats apart
    static constexpr int NBW = kMmNBW;
    static constexpr int kCellBytes = (NC * 20 + 96 + 15) & ~15;   // five per-cell arrays + 24 header ints (flagged cells, group boxes)
    static constexpr int kTabBytes = (NC * TS * 4 + 15) & ~15;
    static constexpr int kF0Cell = (C / 16) * 64 + 16;          // per cell: C/16 x (hi 32 B | lo 32 B) + pad (stride == 4 mod 16 dwords)
    static constexpr int kF0Bytes = NC * kF0Cell;
    static constexpr int kDbufBytes = (NC * DS * 4 + 127) & ~127;
    // NPASS == 1: the D buffer has LDS of its own, so that a block's accumulators leave as soon as its products are done (no
    // barrier, no live range); two-pass shapes (C = 64) keep them across the passes and alias the D buffer over the stage
    static constexpr bool kDbufAlias = NPASS > 1 || R >= 5;  // (large windows: the D buffer alone is 70-90 KB)
    static constexpr int kStage = (kMmLds - kCellBytes - kTabBytes - kF0Bytes - (kDbufAlias ? 0 : kDbufBytes)) & ~127;
    static constexpr int kCap = kStage / SLOT - 32;             // positions that fit (a block may read 31 slots past the region's end)
    static_assert(kDbufBytes <= kStage, "the D buffer may alias the stage");
    static_assert((kStage / 8) < (1 << 18), "slot addresses are packed in 18 bits");
};

// the four-wave shape of local_corr_mw.h (a wave per group: <= kMwRows rows x two column tiles of accumulators)
template <int R, int C>
struct Mw {
    static constexpr int KC = 16, NCH = C / 16, NSUB = 1, NPIECE = 4, SLOT = 64;
    static constexpr int PW = 2 * R + 2, D = 2 * R + 1, K = D * D, TS = 2 * D + 1;
    static constexpr int NC = 64, NW = 4;
    static constexpr int RP = PW + 6;
    static constexpr int DS = ((PW * RP + 31) & ~31) + 5;
    static constexpr int NBW = 2 * kMwRows;
    static constexpr int kCellBytes = (NC * 20 + 96 + 16 + 15) & ~15;   // five per-cell arrays, 24 header ints, a 4-float dump
    static constexpr int kTabBytes = (NC * TS * 4 + 15) & ~15;
    static constexpr int kF0Cell = NCH * 64 + 16;
    static constexpr int kF0Bytes = NC * kF0Cell;
    static constexpr int kDbufBytes = NC * DS * 4;
    static constexpr int kLds = 80 * 1024;
    static constexpr int kStage = (kLds - kCellBytes - kTabBytes - kF0Bytes) & ~127;
    static constexpr int kCap = kStage / SLOT - 32;
};

// swizzle of a slot's 16-byte pieces: piece p of slot s lives at physical piece p ^ mm_swz<KC>(s).  KC = 32 (8 pieces, slot stride
// 32 dwords): (s >> 1) & 7 -- 16 consecutive slots reading one piece pair (p, p ^ 1) in the hardware's b128 lane groups
// {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} touch 16 different 4-bank groups.  KC = 16 (4 pieces, stride 16 dwords): a 2-bit code of
// (s >> 2) & 3 with the same property.
template <int KC>
__device__ __forceinline__ unsigned mm_swz(unsigned s) {
    if constexpr (KC == 32) return (s >> 1) & 7u;
    else return (0x78u >> (((s >> 2) & 3u) * 2u)) & 3u;
}

// MM region geometry from the box: pitch = whole quads, no padding
struct MmRegion {
    int x0, y0, w, h, pitch, nq;
};
__device__ __forceinline__ void mm_region_geometry(MmRegion &u) {
    u.nq = (u.w + 3) >> 2;
    u.pitch = u.nq * 4;
}
// does the region fit the stage (plan launch: C is a run-time value there)
template <int R>
__device__ __forceinline__ bool mm_region_fits_rt(int w, int h, int C) {
    const int pitch = ((w + 3) >> 2) * 4;
#if GFN_MM_DEFAULT == 1
    const int cap = C == 16 ? Mw<R, 16>::kCap : (C == 32 ? Mw<R, 32>::kCap : Mw<R, 64>::kCap);
#else
    const int cap = C == 16 ? Mm<R, 16>::kCap : (C == 32 ? Mm<R, 32>::kCap : Mm<R, 64>::kCap);
#endif
    return (long)pitch * h <= cap && w <= 252 && h <= 255;
}

// ---- staging ---------------------------------------------------------------------------------------------------------------------
// Work item `it` of a pass = quads 16 (it / NSUB) .. + 15 of the region (row major) x the four channel quads of sub-chunk it % NSUB;
// wave w takes items w, w + 16, ...  Lane bits 0-1 and 4-5 = the quad, bits 2-3 = the channel quad (as the lean path: four
// consecutive lanes read 64 contiguous bytes of a plane).  A lane's four loads bring 4 pixels x 4 channels; per pixel they leave as
// 8 bytes of a hi piece and 8 bytes of the matching lo piece.
struct MmItem {
    unsigned voff;   // byte offset of the lane's quad in its first plane (incl. the channel quad's planes)
    unsigned meta;   // bits 0-17: byte address / 8 of the hi piece half of pixels 0-1 (KC = 32: pixels 2-3 sit one piece over: ^ 2);
                     // 18-21: pixels inside the image (CHECK); 22: the lane has a quad; 23: its row lies inside the image (CHECK)
};

// M = the kernel's traits (Mm<R, C> here, Mq<R, C> in local_corr_mq.h): NSUB, KC, SLOT, NPIECE; NW = waves of the workgroup
// OOR: a lane without a quad gets an offset past the descriptor's range (the load returns zeros without touching memory)
template <typename M, int NW, bool CHECK, typename FT, bool OOR = false>
__device__ __forceinline__ MmItem mm_item(const MmRegion &u, int H, int W, int wave, int lane, int k) {
    constexpr unsigned ES = sizeof(FT);
    const int it = wave + NW * k;
    const int qg = it / M::NSUB, sc = it % M::NSUB;  // scalar
    const int cg = sc * 4 + ((lane >> 2) & 3);         // channel quad of the pass
    const int L = qg * 16 + ((lane & 3) | ((lane >> 4) << 2));
    const float inv_nq = __builtin_amdgcn_rcpf((float)max(u.nq, 1));
    int row = (int)(((float)L + 0.5f) * inv_nq);       // L / nq, exact for these sizes (L < 16384)
    int q = L - row * u.nq;
    const bool have = row < u.h;
    if (!have) row = 0, q = 0;
    const int x = u.x0 + 4 * q;
    unsigned xmask = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) xmask |= ((unsigned)(x + j) < (unsigned)W ? 1u : 0u) << j;
    const int gy = u.y0 + row;
    const bool row_in = (unsigned)gy < (unsigned)H;
    MmItem o;
    const int px = CHECK ? (row_in ? gy : 0) * W + max(x, 0) : row * W + x;
    o.voff = (unsigned)px * ES + (unsigned)cg * 4u * (unsigned)(H * W) * ES;
    if (OOR && !have) o.voff = kOffRange;
    const unsigned s0 = (unsigned)(row * u.pitch + 4 * q);           // slot of pixel 0: a multiple of 4
    const unsigned phys = (unsigned)(cg >> 1) ^ mm_swz<M::KC>(s0);   // hi piece of pixels 0-1
    const unsigned a8 = s0 * (M::SLOT / 8) + phys * 2u + (unsigned)(cg & 1);
    o.meta = a8 | (xmask << 18) | (have ? 1u << 22 : 0u) | (row_in ? 1u << 23 : 0u);
    return o;
}

constexpr int kMmPre = 2;  // items of a pass in flight per wave
struct MmLane {
    MmItem it[kMmPre];
};
template <typename FT>
struct MmRegs {
    typename QuadRaw<FT>::type a[kMmPre][4];
};

template <bool CHECK, typename FT, bool OOR = false>
__device__ __forceinline__ void mm_issue(MmRegs<FT> &r, rsrc_t f1r, unsigned pass_off, int H, int W, const MmRegion &u, int ipw, const MmLane &ml) {
    constexpr unsigned ES = sizeof(FT);
    const unsigned plane4 = (unsigned)(H * W) * ES;
    const unsigned so = pass_off + (CHECK ? 0u : (unsigned)(u.y0 * W) * ES);
#pragma unroll
    for (int n = 0; n < kMmPre; ++n) {
        // no branch around the loads (a wave without a second item repeats its first one: L1 hits): behind a branch the loaded
        // registers become phi nodes and the compiler waits for them at the merge -- nothing stays in flight
        const unsigned vo = (n == 0 || n < ipw) ? ml.it[n].voff : (OOR ? kOffRange : ml.it[0].voff);
#pragma unroll
        for (int j = 0; j < 4; ++j) r.a[n][j] = QuadRaw<FT>::load(f1r, vo, so + (unsigned)j * plane4);
    }
}

// one item's 4 pixels x 4 channels -> bf16 hi / lo pieces
template <typename M, bool CHECK, typename FT>
__device__ __forceinline__ void mm_commit_one(unsigned char *stage, const typename QuadRaw<FT>::type (&a)[4], unsigned meta) {
    u32x2_t *s8 = reinterpret_cast<u32x2_t *>(stage);
    const unsigned a01 = meta & 0x3FFFFu;                         // hi piece half of pixels 0-1, in units of 8 bytes
    const unsigned a23 = M::KC == 32 ? a01 ^ 2u : a01;            // pixels 2-3: slot + 2 flips bit 0 of the KC = 32 swizzle
    constexpr unsigned LO = M::NPIECE;                            // lo piece = hi piece ^ (NPIECE / 2) = ^ NPIECE units of 8 bytes
    constexpr unsigned SL = M::SLOT / 8;
    unsigned m = 0xFu;
    if (CHECK) m = ((meta >> 23) & 1u) ? (meta >> 18) & 0xFu : 0u;
    const f32x4 w0 = QuadRaw<FT>::widen(a[0]), w1 = QuadRaw<FT>::widen(a[1]), w2 = QuadRaw<FT>::widen(a[2]), w3 = QuadRaw<FT>::widen(a[3]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool in = !CHECK || ((m >> k) & 1u);
        unsigned h01, l01, h23, l23;
        split_pair(in ? w0[k] : 0.f, in ? w1[k] : 0.f, h01, l01);
        split_pair(in ? w2[k] : 0.f, in ? w3[k] : 0.f, h23, l23);
        const unsigned hi8 = (k < 2 ? a01 : a23) + (unsigned)k * SL;  // k * SL leaves the piece bits alone
        s8[hi8] = u32x2_t{h01, h23};
        s8[hi8 ^ LO] = u32x2_t{l01, l23};
    }
}

template <typename M, bool CHECK, typename FT>
__device__ __forceinline__ void mm_commit(unsigned char *stage, const MmRegs<FT> &r, int ipw, const MmLane &ml) {
#pragma unroll
    for (int n = 0; n < kMmPre; ++n) {
        const unsigned meta = ml.it[n].meta;
        if ((n < ipw) & ((meta >> 22) & 1u)) mm_commit_one<M, CHECK, FT>(stage, r.a[n], meta);
    }
}

// items beyond the kMmPre register-held ones (regions of more than 512 / NSUB quads)
template <typename M, int NW, bool CHECK, typename FT>
__device__ __forceinline__ void mm_rest(unsigned char *stage, rsrc_t f1r, unsigned pass_off, int H, int W, const MmRegion &u, int ipw, int wave, int lane) {
    constexpr unsigned ES = sizeof(FT);
    const unsigned plane4 = (unsigned)(H * W) * ES;
    const unsigned so = pass_off + (CHECK ? 0u : (unsigned)(u.y0 * W) * ES);
    for (int k = kMmPre; k < ipw; ++k) {
        const MmItem it = mm_item<M, NW, CHECK, FT>(u, H, W, wave, lane, k);
        typename QuadRaw<FT>::type a[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = QuadRaw<FT>::load(f1r, it.voff, so + (unsigned)j * plane4);
        if ((it.meta >> 22) & 1u) mm_commit_one<M, CHECK, FT>(stage, a, it.meta);
    }
}

// ---- the kernel -----------------------------------------------------------------------------------------------------------------------
// cell ids are the lean path's: id = half * 32 + row * 8 + column-in-half; group g = id >> 4 = rows 2 (g & 1), 2 (g & 1) + 1 of half
// g >> 1.  D-buffer position of a cell: ((id & 31) << 1) | (id >> 5) -- cells id and id + 32 would otherwise share every bank.
__device__ __forceinline__ int mm_dpos(int cell) { return ((cell & 31) << 1) | (cell >> 5); }

// a tile's loads in flight.  Loop-carried state is kept small on purpose: the scalars of a tile (batch element, tile origin, staging
// region, items per wave) are re-derived from its id and plan where they are needed -- carried along for two tiles they pushed the
// kernel past the scalar register file, uniform values were parked in vector registers and every buffer load became a waterfall loop.
template <int R, int C, typename FT>
struct MmFetch {
    static constexpr int NF0 = C >= 32 ? C / kMmWaves : 2;   // wave w takes channels NF0 w .. (C = 16: waves 0-7 only)
    unsigned wid;          // scalars: tile id, its plan
    int px0, py0, phw, pflags;
    float nx, ny;          // per lane (lane = cell id)
    float f0v[NF0];        // per lane (lane = tile row lane >> 4, column lane & 15)
    MmLane ml;
    MmRegs<FT> pre;
};

struct MmTile {  // the scalars of a tile
    MmRegion u;
    int b, row0, col0, ipw;
    bool valid, interior;
};
template <int R, int C>
__device__ __forceinline__ MmTile mm_tile_scalars(const LcParams &p, unsigned wid, int px0, int py0, int phw, int pflags) {
    MmTile t;
    t.valid = !(pflags & kPlanSecond);     // tiles on the second launch's list are not ours
    t.interior = (pflags & kPlanInterior) != 0;
    t.u.x0 = px0; t.u.y0 = py0; t.u.w = phw & 0xffff; t.u.h = phw >> 16;
    mm_region_geometry(t.u);
    const int tiles = p.tiles_x * p.tiles_y;
    t.b = wid / tiles;
    const int tile = wid - t.b * tiles;
    const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    t.row0 = ty * 4; t.col0 = tx * kTileW;
    const int nitems = ((t.u.h * t.u.nq + 15) >> 4) * Mm<R, C>::NSUB;
    t.ipw = (nitems + kMmWaves - 1) / kMmWaves;
    return t;
}

// issue everything tile `wid` needs from global memory; pl = its plan
template <int R, int C, typename FT>
__device__ __forceinline__ void mm_fetch(const LcParams &p, MmFetch<R, C, FT> &f, unsigned wid, int4 pl, int lane, int wave) {
    typedef MmFetch<R, C, FT> F;
    const int G = p.G, H = p.H, W = p.W;
    const unsigned GG4 = (unsigned)(G * G) * 4u;
    // the plan came through a vector load of a uniform address: tell the compiler it is uniform (scalar offsets of the buffer
    // loads, block-uniform branches -- otherwise every load sits in a waterfall loop)
    f.wid = wid;
    f.px0 = __builtin_amdgcn_readfirstlane(pl.x); f.py0 = __builtin_amdgcn_readfirstlane(pl.y);
    f.phw = __builtin_amdgcn_readfirstlane(pl.z); f.pflags = __builtin_amdgcn_readfirstlane(pl.w);
    const MmTile t = mm_tile_scalars<R, C>(p, wid, f.px0, f.py0, f.phw, f.pflags);
    if (!t.valid) return;  // scalar
    if (wave == 0) {  // scalar: one wave reads the 64 flows and files what the others need (cells, group boxes) in LDS
        const int gi = t.row0 + cell_row(lane), gj = t.col0 + cell_col(lane);
        const bool ok = (gi < G) & (gj < G);
        const rsrc_t flr = make_rsrc(p.flow + (size_t)t.b * 2 * G * G, 2u * GG4);
        const unsigned fo = ok ? (unsigned)(gi * G + gj) * 4u : 0u;
        f.nx = buf_ld(flr, fo, 0u);
        f.ny = buf_ld(flr, fo, GG4);
    }
    {
        const int fr = lane >> 4, fc = lane & 15;
        const bool fok = (t.row0 + fr < G) & (t.col0 + fc < G);
        const unsigned fgoff = fok ? (unsigned)((t.row0 + fr) * G + t.col0 + fc) * 4u : 0u;
        const rsrc_t f0r = make_rsrc(p.f0 + (size_t)t.b * p.f0_bs, (unsigned)C * GG4);
        const int fw = wave * F::NF0 < C ? wave : 0;  // idle waves (C = 16) repeat wave 0's loads: no branch around loads
#pragma unroll
        for (int k = 0; k < F::NF0; ++k) f.f0v[k] = buf_ld(f0r, fgoff, (unsigned)(fw * F::NF0 + k) * GG4);
    }
    const rsrc_t f1r = make_rsrc(f1_of<FT>(p, t.b), (unsigned)C * (unsigned)(H * W) * (unsigned)sizeof(FT));
    // one code path for interior and border tiles (out-of-image pixels are masked at the commit): a block-uniform branch around the
    // loads turns the loaded registers into phi nodes, which the compiler parks in scratch
#pragma unroll
    for (int n = 0; n < kMmPre; ++n) f.ml.it[n] = mm_item<Mm<R, C>, kMmWaves, true, FT>(t.u, H, W, wave, lane, n);
    mm_issue<true, FT>(f.pre, f1r, 0u, H, W, t.u, t.ipw, f.ml);
}

template <int R, int C, typename FT>
__global__ __launch_bounds__(kMmThreads, 4) void local_corr_mm1_kernel(LcParams p) {
    typedef Mm<R, C> M;
    typedef MmFetch<R, C, FT> F;
    constexpr int PW = M::PW, D = M::D, K = M::K, TS = M::TS, NC = M::NC, RP = M::RP, DS = M::DS, NBW = M::NBW;
    constexpr int NSUB = M::NSUB, NPASS = M::NPASS, KC = M::KC, NF0 = F::NF0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *dbuf = reinterpret_cast<float *>(smem + (M::kDbufAlias ? 0 : M::kStage));
    unsigned char *misc = smem + M::kStage + (M::kDbufAlias ? 0 : M::kDbufBytes);
    int *cellX0 = reinterpret_cast<int *>(misc);
    int *cellY0 = cellX0 + NC;
    float *cellNx = reinterpret_cast<float *>(cellY0 + NC);
    float *cellNy = cellNx + NC;
    int *cellFlag = reinterpret_cast<int *>(cellNy + NC);
    int *hdr = cellFlag + NC;
    float *tab = reinterpret_cast<float *>(misc + M::kCellBytes);
    unsigned char *f0b = misc + M::kCellBytes + M::kTabBytes;

    const int lane0 = threadIdx.x & 63;
    const int wave0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int G = p.G, H = p.H, W = p.W;
    const float xhi = p.win_xhi, xlo = -xhi, yhi = p.win_yhi, ylo = -yhi;
    const unsigned GG4 = (unsigned)(G * G) * 4u;
    const unsigned total = (unsigned)(p.B * p.tiles_x * p.tiles_y);
    const int4 *plans = reinterpret_cast<const int4 *>(p.plan);

    // tiles bid, bid + gridDim, ... of the XCD-contiguous order: at any time an XCD's CUs work on neighbouring tiles
    unsigned v = blockIdx.x;
    if (v >= total) return;
    F cur;
    {
        const unsigned wid = gfn::xcd_remap(v, total);
        mm_fetch<R, C, FT>(p, cur, wid, plans[kPlanV4 * wid], lane0, wave0);
    }
    for (;;) {
        // the lane and wave numbers are made opaque per iteration: the compiler otherwise hoists every lane-derived constant of the
        // tile body (item geometry, cell coordinates, operand offsets) out of the loop and spills them around it
        int lane = lane0, wave = wave0;
        asm volatile("" : "+v"(lane), "+s"(wave));
        const int tid = wave * 64 + lane;
        const unsigned vn = v + gridDim.x;
        const bool has_next = vn < total;
        const unsigned wid_next = has_next ? gfn::xcd_remap(vn, total) : 0u;
        const int4 pl_next = plans[kPlanV4 * wid_next];   // consumed behind the first barrier: a whole staging phase to arrive
        F nxt;
        nxt.pflags = kPlanSecond;
        const MmTile ct = mm_tile_scalars<R, C>(p, cur.wid, cur.px0, cur.py0, cur.phw, cur.pflags);
        if (ct.valid) {  // scalar
#ifdef GFN_ABLATE
            const bool stamping = ABL(p, 512) && (blockIdx.x % 97) == 48 && v >= 3 * gridDim.x && v < 4 * gridDim.x && (tid & 63) == 0 && (tid >> 6) < 2;
            long long stamp[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
            STAMP(0);
            const MmRegion u = ct.u;
            const int b = ct.b, row0 = ct.row0, col0 = ct.col0, ipw = ct.ipw;
            const rsrc_t f1r = make_rsrc(f1_of<FT>(p, b), (unsigned)C * (unsigned)(H * W) * (unsigned)sizeof(FT));
            // ---- per-cell set-up (wave 0), f0 block and the staged pixels -> LDS ------------------------------------------------------
            if (wave == 0) {  // scalar
                const int my_gi = row0 + cell_row(lane), my_gj = col0 + cell_col(lane);
                const bool my_ok = (my_gi < G) & (my_gj < G);
                const float my_nx = my_ok ? cur.nx : 0.f, my_ny = my_ok ? cur.ny : 0.f;
                const CellBox c = cell_box<PW>(my_ok, my_nx, my_ny, xlo, ylo, W, H);
                cellX0[lane] = c.X0;
                cellY0[lane] = c.Y0;
                cellNx[lane] = my_nx;
                cellNy[lane] = my_ny;
                cellFlag[lane] = c.flag;
                const unsigned long long slow_mask = __ballot(c.flag == kCellSlow);
                if (lane == 0) hdr[4] = __popcll(slow_mask);
                // boxes of the four groups' windows (one DPP row each): x0, y0, y1 into hdr[8 + 4 g ..]
                const int rx0 = row_min_i32(c.bx0), ry0 = row_min_i32(c.by0), ry1 = row_min_i32(-c.by1);
                if ((lane & 15) == 15) {
                    int *gb = hdr + 8 + (lane >> 4) * 4;
                    gb[0] = rx0; gb[1] = ry0; gb[2] = -ry1;
                }
            }
            if (wave * NF0 < C) {  // scalar
                // channels NF0 wave .. of cell fcell (inside one 16-channel sub-chunk): NF0 bf16 into the hi half of the cell's slot
                // of that sub-chunk, NF0 into the lo half
                const int fr = lane >> 4, fc = lane & 15;
                const bool fok = (row0 + fr < G) & (col0 + fc < G);
                const int fcell = ((fc >> 3) << 5) | (fr << 3) | (fc & 7);
                const int ch0 = wave * NF0;  // scalar
                unsigned hi[NF0 / 2], lo[NF0 / 2];
#pragma unroll
                for (int k = 0; k < NF0 / 2; ++k) split_pair(fok ? cur.f0v[2 * k] : 0.f, fok ? cur.f0v[2 * k + 1] : 0.f, hi[k], lo[k]);
                unsigned *slot = reinterpret_cast<unsigned *>(f0b + fcell * M::kF0Cell + (ch0 >> 4) * 64 + (ch0 & 15) * 2);
                if constexpr (NF0 == 2) {
                    slot[0] = hi[0]; slot[8] = lo[0];
                } else {
                    static_assert(NF0 == 2 || NF0 == 4, "C is 16, 32 or 64");
                    *reinterpret_cast<u32x2_t *>(slot) = u32x2_t{hi[0], hi[NF0 / 2 - 1]};
                    *reinterpret_cast<u32x2_t *>(slot + 8) = u32x2_t{lo[0], lo[NF0 / 2 - 1]};
                }
            }
            STAMP(1);
            mm_commit<M, true, FT>(smem, cur.pre, ipw, cur.ml);
            mm_rest<M, kMmWaves, true, FT>(smem, f1r, 0u, H, W, u, ipw, wave, lane);
            STAMP(2);
            __syncthreads();
            STAMP(3);
            // fraction table: the reference's fp32 coordinate of every tap column / row of every cell (lane = cell, wave = tap index);
            // read by the epilogue two barriers from here
            {
                const float cnx = cellNx[lane], cny = cellNy[lane];
                const int cX0 = cellX0[lane], cY0 = cellY0[lane];
                bool tab_bad = false;
                constexpr int NTAB = (2 * D + kMmWaves - 1) / kMmWaves;
#pragma unroll
                for (int n = 0; n < NTAB; ++n) {
                    const int a = wave + n * kMmWaves;   // scalar
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
                if (tab_bad && atomicOr(&cellFlag[lane], kCellSlow) == 0) atomicAdd(&hdr[4], 1);  // rare
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
                nb = any ? min((by1 - by0 - rp + 1) >> 1, NBW) : 0;  // the plan guarantees <= NBW
            }

            // ---- the next tile's loads go out now and stay in flight until the top of the next iteration (shapes that keep their
            // accumulators across passes: behind the accumulators, see below) ------------------------------------------------------
            if (!M::kDbufAlias && has_next) mm_fetch<R, C, FT>(p, nxt, wid_next, pl_next, lane, wave);
            STAMP(4);

            // ---- products --------------------------------------------------------------------------------------------------------
            // lane = position m (lane & 15) of the block + 16 * k-group q: A = the piece pair of sub-chunk sc of slot (row, gx0 + m);
            // q = 0, 1: hi octets 2 sc, 2 sc + 1; q = 2, 3: the lo octets.  Afterwards the lane holds, per block, four consecutive
            // positions (columns gx0 + 4 q ..) of region row gy0 + 2 i for cell g * 16 + m.
            const int mq = lane >> 4, mm = lane & 15;
            const unsigned slot0 = (unsigned)(gy0 * u.pitch + gx0 + mm);
            const unsigned sstep = (unsigned)(2 * u.pitch);
            const unsigned lpiece = (unsigned)((mq >> 1) * (M::NPIECE / 2) + (mq & 1));   // logical piece of sub-chunk 0
            const unsigned b_addr = (unsigned)((g * 16 + mm) * M::kF0Cell + (mq & 1) * 16);
            const int cellg = g * 16 + mm;
            float *dwin;      // the lane's first value of block 0 in the D buffer
            bool col_ok;
            int dy0;
            {
                const int X0 = cellX0[cellg], Y0 = cellY0[cellg];
                const bool has = X0 != kFar;                                  // false: off the grid, flagged, or its window misses the image
                const int dx0 = has ? gx0 + 4 * mq - (X0 - u.x0) : -1000;    // window column of the first of the four
                dy0 = has ? gy0 - (Y0 - u.y0) : 0;                           // window row of block 0
                col_ok = (unsigned)(dx0 + 3) < (unsigned)(PW + 3);
                dwin = dbuf + mm_dpos(cellg) * DS + 3 + dy0 * RP + dx0;
            }
            auto extract = [&](int i, const f32x4 &a) {
                if (col_ok & ((unsigned)(dy0 + 2 * i) < (unsigned)PW)) {
                    dwin[2 * i * RP + 0] = a[0];
                    dwin[2 * i * RP + 1] = a[1];
                    dwin[2 * i * RP + 2] = a[2];
                    dwin[2 * i * RP + 3] = a[3];
                }
            };
            auto a_operand = [&](int i, int sc) {
                const unsigned s = slot0 + (unsigned)i * sstep;
                const unsigned base = s * (unsigned)M::SLOT + ((lpiece ^ mm_swz<KC>(s)) << 4);
                return *reinterpret_cast<const bf16x8_t *>(smem + (base ^ (unsigned)(sc * 32)));
            };
            if constexpr (!M::kDbufAlias) {
                // Two blocks at a time, complete over the channels, filed in the D buffer at once: 8 accumulator registers alive (16
                // with the next pair's products under way) instead of 4 NBW.  Rows past the group's box repeat its last row into
                // values nobody files (they fail every cell's row test).
                bf16x8_t b1[NSUB], b2[NSUB];
#pragma unroll
                for (int sc = 0; sc < NSUB; ++sc) {
                    b1[sc] = *reinterpret_cast<const bf16x8_t *>(f0b + b_addr + sc * 64);
                    b2[sc] = *reinterpret_cast<const bf16x8_t *>(f0b + b_addr + sc * 64 + 32);
                }
#pragma unroll
                for (int i0 = 0; i0 < NBW; i0 += 2) {
                    if (i0 < nb) {  // scalar
                        bf16x8_t a[2][NSUB];
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int sc = 0; sc < NSUB; ++sc) a[j][sc] = a_operand(min(i0 + j, nb - 1), sc);
                        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                        for (int sc = 0; sc < NSUB; ++sc) {
#pragma unroll
                            for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j][sc], b1[sc], acc[j], 0, 0, 0);
#pragma unroll
                            for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j][sc], b2[sc], acc[j], 0, 0, 0);
                        }
                        extract(i0, acc[0]);
                        extract(i0 + 1, acc[1]);
                    }
                }
                STAMP(5);
            } else {
                // two passes over the channels (C = 64): the accumulators live across them, the D buffer aliases the stage
                f32x4 acc[NBW];
#pragma unroll
                for (int i = 0; i < NBW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps) {
                    const bool more = ps + 1 < NPASS;
                    const unsigned next_off = (unsigned)((ps + 1) * KC) * (unsigned)(H * W) * (unsigned)sizeof(FT);
                    MmRegs<FT> pre2;
                    if (more) mm_issue<true, FT>(pre2, f1r, next_off, H, W, u, ipw, cur.ml);  // next pass's loads: in flight across the products
#pragma unroll
                    for (int sc = 0; sc < NSUB; ++sc) {
                        const bf16x8_t b1 = *reinterpret_cast<const bf16x8_t *>(f0b + b_addr + (ps * NSUB + sc) * 64);
                        const bf16x8_t b2 = *reinterpret_cast<const bf16x8_t *>(f0b + b_addr + (ps * NSUB + sc) * 64 + 32);
                        // the operand addresses are re-derived per sub-chunk (5 instructions a block): kept in common across the
                        // sub-chunks and passes they cost NBW registers beside 4 NBW accumulators and 32 of loads in flight
                        unsigned slot_sc = slot0;
                        asm volatile("" : "+v"(slot_sc));
                        auto a_op = [&](int i) {
                            const unsigned s = slot_sc + (unsigned)i * sstep;
                            const unsigned base = s * (unsigned)M::SLOT + ((lpiece ^ mm_swz<KC>(s)) << 4);
                            return *reinterpret_cast<const bf16x8_t *>(smem + (base ^ (unsigned)(sc * 32)));
                        };
#pragma unroll
                        for (int i0 = 0; i0 < NBW; i0 += 2) {
                            if (i0 < nb) {  // scalar
                                bf16x8_t a[2];
#pragma unroll
                                for (int j = 0; j < 2; ++j) a[j] = a_op(min(i0 + j, nb - 1));
#pragma unroll
                                for (int j = 0; j < 2; ++j) acc[i0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], b1, acc[i0 + j], 0, 0, 0);
#pragma unroll
                                for (int j = 0; j < 2; ++j) acc[i0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], b2, acc[i0 + j], 0, 0, 0);
                            }
                        }
                    }
                    if (more) {
                        __syncthreads();  // everyone is done reading this pass's pixels
                        mm_commit<M, true, FT>(smem, pre2, ipw, cur.ml);
                        mm_rest<M, kMmWaves, true, FT>(smem, f1r, next_off, H, W, u, ipw, wave, lane);
                        __syncthreads();
                    }
                }
                STAMP(5);
                __syncthreads();  // the D buffer aliases the stage
#pragma unroll
                for (int i = 0; i < NBW; ++i)
                    if ((i & ~1) < nb) extract(i, acc[i]);
                // the next tile's loads go out once the accumulators are gone (their 4 NBW registers and the two passes' load
                // registers leave no room earlier) and fly under the epilogue -- (2 r + 1)^2 stores per cell at these radii
                if (has_next) mm_fetch<R, C, FT>(p, nxt, wid_next, pl_next, lane, wave);
            }
            STAMP(6);
            __syncthreads();
            STAMP(7);
            {
                // lane -> cell so that a wave stores whole 64-byte grid-row segments; wave = tap row ky
                const int er = lane >> 4, ec = lane & 15;
                const int cell = ((ec >> 3) << 5) | (er << 3) | (ec & 7);
                const int gi = row0 + er, gj = col0 + ec;
                const int flag = cellFlag[cell];
                if ((gi < G) & (gj < G) & !(flag & kCellSlow)) {
                    const bool empty = (flag & kCellEmpty) != 0;
                    const float *dc = dbuf + mm_dpos(cell) * DS + 3;
                    const float *tc = tab + cell * TS;
                    const unsigned goff = (unsigned)(gi * G + gj) * 4u;
                    const rsrc_t outr = make_rsrc(p.out + (size_t)b * p.out_bs, (unsigned)K * GG4);
                    constexpr int NR = (D + kMmWaves - 1) / kMmWaves;
#pragma unroll
                    for (int n = 0; n < NR; ++n) {
                        const int ky = wave + n * kMmWaves;  // scalar
                        if (ky < D) {
                            const float wy1 = tc[D + ky];
                            const float wy1s = wy1 * p.inv_sqrt_c, wy0s = (1.f - wy1) * p.inv_sqrt_c;
                            const float *dd = dc + ky * RP;
                            float m[PW], val[D];
#pragma unroll
                            for (int x = 0; x < PW; ++x) m[x] = fmaf(dd[RP + x], wy1s, dd[x] * wy0s);
#pragma unroll
                            for (int kx = 0; kx < D; ++kx) {
                                const float wx1 = tc[kx];
                                val[kx] = fmaf(m[kx + 1], wx1, m[kx] * (1.f - wx1));
                            }
                            // The next tile's loads are still counted in vmcnt.  Loads and stores retire out of order with respect
                            // to each other on gfx9, so once a store is pending the compiler's wait in front of the first use of a
                            // loaded register is vmcnt(0): it would sit at the top of the next tile and drain these stores (2.6 k cycles
                            // measured).  Waiting for the loads HERE, before the first store, costs nothing (they were issued a
                            // products phase ago) and leaves the stores to retire under the next tile's set-up.
                            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt / lgkmcnt untouched
#pragma unroll
                            for (int kx = 0; kx < D; ++kx) buf_st_nt(outr, goff, (unsigned)(ky * D + kx) * GG4, empty ? 0.f : val[kx]);
                        }
                    }
                }
            }
            STAMP(8);
            // ---- flagged cells: general per-tap routine (about one cell in 10^4) -------------------------------------------------
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
                for (int e = tid; e < totalk; e += kMmThreads) {
                    const int cell = cellX0[e / K], k = e % K;
                    const int gi = row0 + cell_row(cell), gj = col0 + cell_col(cell);
                    p.out[(size_t)b * p.out_bs + ((size_t)k * G + gi) * G + gj] =
                        tap_general<FT>(p, b, gi, gj, k / D, k % D, D, cellNx[cell], cellNy[cell]);
                }
            }
            __syncthreads();  // cells, table, f0 block and D buffer are the next tile's from here on
            STAMP(9);
#ifdef GFN_ABLATE
            if (stamping)
                printf("mm1 r%d wave %d (cycles from the top of the tile): set-up %lld | committed %lld | barrier %lld | next tile's loads issued %lld | "
                       "products %lld | last blocks filed %lld | barrier %lld | stores issued %lld | barrier %lld\n",
                       R, tid >> 6, stamp[1] - stamp[0], stamp[2] - stamp[0], stamp[3] - stamp[0], stamp[4] - stamp[0], stamp[5] - stamp[0],
                       stamp[6] - stamp[0], stamp[7] - stamp[0], stamp[8] - stamp[0], stamp[9] - stamp[0]);
#endif
        } else if (has_next) {
            mm_fetch<R, C, FT>(p, nxt, wid_next, pl_next, lane, wave);
        }
        if (!has_next) break;
        cur = nxt;
        v = vn;
    }
}
