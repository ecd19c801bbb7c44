// local_corr_mstage.h -- split-bf16 operands and quad staging for the matrix-core local-correlation kernel (local_corr_mq.h, r >= 5 on
// 64-channel maps).  Included by local_corr.hip after local_corr_lean.h (same namespace: cell boxes, buffer addressing, the fraction
// table and the epilogue arithmetic are the lean path's).
//
// The product: D[cell][position] = sum_c f0[cell][c] * f1[c][position] is (positions x channels) . (channels x cells).  fp32 accuracy
// from bf16 operands: every value is split x = hi + lo when it is filed in LDS (both pieces round-to-nearest bf16: residual <=
// 2^-18 |x|), and the K = 32 of v_mfma_f32_16x16x32_bf16 holds 16 channels as [hi | lo]:
//      A (positions) = [f1_hi(16) | f1_lo(16)],   B1 (cells) = [f0_hi | f0_hi],   B2 = [f0_lo | f0_lo]
//      mfma(A, B1) + mfma(A, B2) = sum_c (f1_hi + f1_lo)(f0_hi + f0_lo):  all four partial products, fp32 accumulation,
// a product off by <= 2^-17 relative (tests: 1e-4 * max(1, |ref|) against the oracle; measured a few 1e-6).  fp16 maps split exactly.
//
// History: these helpers were written in round 3 for matrix-core kernels of the r = 3 / 4 shapes (a persistent 16-wave workgroup per
// CU, then four-wave workgroups with a wave per 2 x 8-cell group).  Both were parity-green and 10-25 % slower than the fp32 FMA
// kernel of local_corr_lean.h (profiles/r03_local_corr_mm.md: a quarter of a box's products are useful, and the bf16 split plus the
// filing of the accumulators put back the vector instructions the FMAs freed); they stayed parked behind a build flag through round
// 4 and were deleted in round 5 (git history: csrc/local_corr_mm.h, csrc/local_corr_mw.h at e0234d8).
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

// M = the kernel's traits (Mq<R, C> in local_corr_mq.h): NSUB, KC, SLOT, NPIECE; NW = waves of the workgroup
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
