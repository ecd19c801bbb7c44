// local_corr_stage.h -- staging of f1 into the LDS stage by 16-byte quads through buffer descriptors.  Included by local_corr.hip in
// front of the round-1 tile routine (r >= 5 stages this way since round 3) and of local_corr_lean.h (whose tile kernels it was
// written for in round 2).  Shares kSlotV4 / kWaves with local_corr.hip.

// region geometry from the box, identically in the plan kernel and the tile kernel
struct RowPlan {  // block-uniform (scalars)
    int x0, y0, w, h, pitch;
    int nq;       // 16-byte quads per region row
    int nitems;   // work items of a 16-channel chunk (16 consecutive quads of the region, row major, x 4 channel quads): a
                  // multiple of 8, nitems / 8 per wave
};

// ---- buffer addressing --------------------------------------------------------------------------------------------
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 make_i32x4(int a, int b, int c, int d) { i32x4 v = {a, b, c, d}; return v; }
__device__ __forceinline__ rsrc_t make_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);  // raw buffer, 32-bit data format
}
__device__ __forceinline__ float buf_ld(rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ f32x4 buf_ld4(rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f16x4 buf_ld_h4(rsrc_t r, unsigned voff, unsigned soff) {
    // the builtin's result is bit-cast as a whole: picking the two dwords apart (.x / .y of a 2-vector) made hipcc 7.2 narrow the
    // load to one dword and reuse it for both halves
    return __builtin_bit_cast(f16x4, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
}
// a quad of 4 pixels of one channel plane as it comes from memory, and widened to fp32
template <typename FT> struct QuadRaw;
template <> struct QuadRaw<float> {
    typedef f32x4 type;
    static __device__ __forceinline__ type load(rsrc_t r, unsigned voff, unsigned soff) { return buf_ld4(r, voff, soff); }
    static __device__ __forceinline__ f32x4 widen(type v) { return v; }
};
template <> struct QuadRaw<_Float16> {
    typedef f16x4 type;
    static __device__ __forceinline__ type load(rsrc_t r, unsigned voff, unsigned soff) { return buf_ld_h4(r, voff, soff); }
    static __device__ __forceinline__ f32x4 widen(type v) {
        f32x4 o = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
        return o;
    }
};
#ifndef GFN_LEAN_ST_AUX
#define GFN_LEAN_ST_AUX 2  // 2 = nt (streaming store), 0 = plain
#endif
__device__ __forceinline__ void buf_st_nt(rsrc_t r, unsigned voff, unsigned soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, (int)voff, (int)soff, GFN_LEAN_ST_AUX);
}

// ---- staging: 16-byte quads along the row --------------------------------------------------------------------------
// Work item it (of a 16-channel chunk) = quads 16 it .. 16 it + 15 of the region in row-major order, all four channel quads;
// wave w takes items w, w + 8, ...  Lane bits 0-1 and 4-5 = the quad, bits 2-3 = the channel quad cg: four loads (one per
// channel plane of the channel quad) bring 4 pixels x 4 channels, which leave as four 16-byte slot writes (pixel-major
// [pixel][16 channels + pad]).  The bank of a slot write is 4 * quad + 5 * pixel + cg (mod 16, in 16-byte units): sixteen
// lanes of one channel quad hit only four bank groups (measured with all 64 lanes on one channel quad: 30 % of the LDS-active
// cycles of the tile kernel were bank conflicts, all of them these writes); four consecutive quads x four channel quads hit
// sixteen.  Four consecutive lanes still read 64 contiguous bytes of one plane (with cg on the lowest lane bits the texture
// path saw four cache lines per lane quad and the op went from 105 to 120 us).
template <int N, typename FT>
struct QuadRegs {
    typename QuadRaw<FT>::type a[N][4];  // [item][channel of the quad] -> 4 pixels, as loaded (fp16 is widened at the commit)
};

constexpr unsigned kOffRange = 0x7FFFFFF0u;  // a voffset no descriptor of ours covers (planes are < 2^30 bytes)
constexpr int kQuadPre = 2;  // work items of a chunk in flight per wave (regions needing more per wave finish them in a loop)
struct QuadItem {
    unsigned voff;        // byte offset of the lane's quad: (row * W + x) elements + the channel quad's four planes
    unsigned meta;        // bits 0-12: index of the lane's first pixel slot (+ cg) in units of 80 / UNIT bytes (UNIT 5: float4s, the
                          // fp32 stage; UNIT 10: 8-byte pieces, the bf16 hi/lo stage of local_corr_mstage.h); 13-16: pixels of the quad
                          // inside the image (CHECK); 17: the lane has a quad in this item; 18: its row lies inside the image (CHECK)
};
struct QuadLane {         // per lane and region, the first kQuadPre items of this wave
    QuadItem it[kQuadPre];
};

// item k of wave `wave`: where the lane's quad comes from and where it goes
template <bool CHECK, typename FT, int UNIT = kSlotV4, bool OOR = false>
__device__ __forceinline__ QuadItem quad_item(const RowPlan &u, int H, int W, int wave, int lane, int k) {
    constexpr unsigned ES = sizeof(FT);
    const int cg = (lane >> 2) & 3;
    const int L = (wave + 8 * k) * 16 + ((lane & 3) | ((lane >> 4) << 2));  // quad of the region, row major
    const float inv_nq = __builtin_amdgcn_rcpf((float)max(u.nq, 1));
    int row = (int)(((float)L + 0.5f) * inv_nq);      // L / nq, exact for these sizes (L < 1024)
    int q = L - row * u.nq;
    const bool have = row < u.h;
    if (!have) row = 0, q = 0;                        // idle lanes: the slot arithmetic below stays in range, the load is switched off
    const int x = u.x0 + 4 * q;
    unsigned xmask = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) xmask |= ((unsigned)(x + j) < (unsigned)W ? 1u : 0u) << j;
    const int gy = u.y0 + row;
    const bool row_in = (unsigned)gy < (unsigned)H;
    QuadItem o;
    // border tiles: x0 is a multiple of 4 (plan launch), so a quad never straddles the left image edge; quads left of the image
    // and rows outside it point at a pixel inside and are zeroed at the commit, like the pixels that hang over the right edge
    const int px = CHECK ? (row_in ? gy : 0) * W + max(x, 0) : row * W + x;  // !CHECK: relative to the region's first row
    // idle lanes ask for an offset past the descriptor's range: the buffer load returns zeros without a memory access (a repeated
    // valid address cost a full trip through the texture-address path, which is what the tile kernels queue for)
    // (OOR: the r >= 3 kernels; the r <= 2 kernels live on 80 registers and the extra select spills: they keep the repeated address)
    o.voff = (have || !OOR) ? (unsigned)px * ES + (unsigned)cg * 4u * (unsigned)(H * W) * ES : kOffRange;
    o.meta = (unsigned)((row * u.pitch + 4 * q) * UNIT + cg) | (xmask << 13) | (have ? 1u << 17 : 0u) | (row_in ? 1u << 18 : 0u);
    return o;
}

template <int N, bool CHECK, typename FT, int UNIT = kSlotV4, bool OOR = false>
__device__ __forceinline__ void quad_issue(QuadRegs<N, FT> &r, rsrc_t f1r, unsigned chunk_off, int H, int W, const RowPlan &u, int wave,
                                           int lane, const QuadLane &ql, int k0) {
    constexpr unsigned ES = sizeof(FT);
    const unsigned plane4 = (unsigned)(H * W) * ES;  // bytes of a channel plane
    const int ipw = u.nitems >> 3;  // items per wave
    // every staged row of an unchecked tile lies inside the image: its first row goes into the scalar offset
    const unsigned so = chunk_off + (CHECK ? 0u : (unsigned)(u.y0 * W) * ES);
#pragma unroll
    for (int n = 0; n < N; ++n) {
        // no branch around the loads: an item past the wave's last one repeats the last one (L1 hits, result unused)
        unsigned vo;
        if (k0 == 0 && n < kQuadPre) vo = (n == 0 || n < ipw) ? ql.it[n].voff : (OOR ? kOffRange : ql.it[0].voff);
        else vo = quad_item<CHECK, FT, UNIT, OOR>(u, H, W, wave, lane, max(min(k0 + n, ipw - 1), 0)).voff;
        // a wave whose item n lies wholly past the region's last quad issues nothing for it (wave-uniform branch)
        const bool real = !OOR || n == 0 || (wave + 8 * (k0 + n)) * 16 < u.h * u.nq;
        if (real) {
#pragma unroll
            for (int j = 0; j < 4; ++j) r.a[n][j] = QuadRaw<FT>::load(f1r, vo, so + (unsigned)j * plane4);
        }
    }
}

template <int N, bool CHECK, typename FT>
__device__ __forceinline__ void quad_commit(float4 *s4, const QuadRegs<N, FT> &r, int H, int W, const RowPlan &u, int wave, int lane,
                                            const QuadLane &ql, int k0) {
    const int ipw = u.nitems >> 3;
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const unsigned meta = (k0 == 0 && n < kQuadPre) ? ql.it[n].meta : quad_item<CHECK, FT>(u, H, W, wave, lane, k0 + n).meta;
        if ((k0 + n < ipw) & ((meta >> 17) & 1u)) {
            float4 *dst = s4 + (meta & 0x1FFFu);
            unsigned m = 0xFu;
            if (CHECK) m = ((meta >> 18) & 1u) ? (meta >> 13) & 0xFu : 0u;
            const f32x4 w0 = QuadRaw<FT>::widen(r.a[n][0]), w1 = QuadRaw<FT>::widen(r.a[n][1]), w2 = QuadRaw<FT>::widen(r.a[n][2]),
                        w3 = QuadRaw<FT>::widen(r.a[n][3]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float4 v = make_float4(w0[k], w1[k], w2[k], w3[k]);
                if (CHECK && !((m >> k) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                dst[k * kSlotV4] = v;
            }
        }
    }
}

template <bool CHECK, typename FT>
__device__ __forceinline__ void quad_rest(float4 *s4, rsrc_t f1r, unsigned chunk_off, int H, int W, const RowPlan &u, int wave, int lane,
                                          const QuadLane &ql, int done) {
    for (int k0 = done; k0 < (u.nitems >> 3); ++k0) {  // only regions of more than 256 quads (rare)
        QuadRegs<1, FT> r;
        quad_issue<1, CHECK, FT>(r, f1r, chunk_off, H, W, u, wave, lane, ql, k0);
        quad_commit<1, CHECK, FT>(s4, r, H, W, u, wave, lane, ql, k0);
    }
}

