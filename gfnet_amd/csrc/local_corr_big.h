// local_corr_big.h -- round 5: the lean fp32 tile path on 8 x 16-cell tiles ("two stacked 4 x 16 sub-tiles per
// workgroup sharing plan, region and barrier sequence").  Included by local_corr.hip behind local_corr_lean.h (same namespace: plan
// records, cell boxes, fraction arithmetic, buffer addressing).
//
// Why: the stage ablation of the 4 x 16 kernel (profiles/r05_ablate_lean.txt) puts 30 % of an r = 4 call (39 % at r = 2) in the
// "skeleton" -- what a workgroup does whatever its windows hold: dispatch, plan decode, flows, cell set-up, per-wave geometry, six
// barriers -- and another 22 % (30 %) in staging a region that is mostly halo: 17 region rows for 4 rows of cells.  A workgroup that
// takes TWO vertically adjacent tiles pays the skeleton once for 128 cells and stages 24 rows instead of 2 x 17.
//
// What changes against lean_tile (results are bit-identical: same products, same fmaf order over the channels, same epilogue):
//   * 8-channel chunks (48-byte slots = 8 channels + 16 bytes of pad: b128 reads of 16 consecutive slots stay conflict-free), so
//     that the taller region still fits: 24 x 42 slots = 48 KB; the D buffer of 128 cells (52 KB) aliases it;
//   * a staging item is 32 quads x 2 channel quads (lane bits 0-1 and 3-5: the quad, bit 2: the channel quad); a chunk of the
//     typical region is 8-9 items: ONE per wave (the 4 x 16 kernel padded 11 to 16);
//   * four D-stage rounds per wave (two per sub-tile), 28 accumulators + 28 slot addresses per lane;
//   * the f0 block goes through LDS one 8-channel chunk at a time (6 KB);
//   * no plan change: the workgroup reads the plan records of its two tiles and takes this path when both are whole-tile interior
//     tiles whose joint region fits; any other pair (border tiles, halves, listed tiles, a single last row) runs lean_tile twice.

#ifndef GFN_LEAN_BIG
#define GFN_LEAN_BIG 0
#endif

template <int R>
struct Big {
    static constexpr int PW = 2 * R + 2, P = PW * PW, NP = (P + 15) / 16, D = 2 * R + 1, K = D * D, TS = 2 * D + 1;
    static constexpr int NC = 128, DS = P + 1, kSkew = 16, CS = 12;
    static constexpr int kSlotB = 3;  // float4s per staged pixel: 8 channels + 4 floats of pad
    static constexpr int kDbuf = ((NC * DS + 3 * kSkew) * 4 + 15) & ~15;   // rows of the cells 32 q .. 32 q + 31 are skewed by 16 q floats (bank spread, lean_tile)
    static constexpr int kStageMin = R <= 2 ? 36 * 1024 : 52 * 1024;
    static constexpr int kStage = kDbuf > kStageMin ? ((kDbuf + 1023) & ~1023) : kStageMin;
    static constexpr int kCap = kStage / (kSlotB * 16);
    static constexpr int kCellBytes = (NC * 20 + 32 + 15) & ~15;
    static constexpr int kTabBytes = ((NC * TS + 3 * kSkew) * 4 + 15) & ~15;
    static constexpr int kF0Bytes = NC * CS * 4;
    static constexpr int kLds = kStage + kCellBytes + kTabBytes + kF0Bytes;
    static constexpr bool kOn = GFN_LEAN_BIG != 0 && (R == 4 || R == 3);
};

template <int R>
__device__ __forceinline__ bool big_region_fits(RowPlan &u) {
    constexpr int PW = Big<R>::PW;
    u.nq = (u.w + 3) >> 2;
    const int w4 = u.nq * 4;
    u.pitch = w4 + ((PW - w4) & 15);  // pitch == patch width (mod 16): conflict-free b128 reads across patch rows
    if (u.pitch * u.h > Big<R>::kCap || u.w > 64 || u.w <= 0 || u.h <= 0) return false;
    u.nitems = (u.h * u.nq + 31) >> 5;  // items of 32 quads
    return true;
}

struct BigItem {
    unsigned voff;  // byte offset of the lane's quad in its first plane (kOffRange: no quad)
    unsigned meta;  // bits 0-12: float4 index of the lane's first slot (+ channel quad); bit 17: the lane has a quad
};

// item k of wave `wave` (interior regions only: every staged pixel lies inside the image)
template <typename FT>
__device__ __forceinline__ BigItem big_item(const RowPlan &u, int H, int W, int wave, int lane, int k) {
    constexpr unsigned ES = sizeof(FT);
    const int cg = (lane >> 2) & 1;
    const int L = (wave + 8 * k) * 32 + ((lane & 3) | ((lane >> 3) << 2));  // quad of the region, row major
    const float inv_nq = __builtin_amdgcn_rcpf((float)max(u.nq, 1));
    int row = (int)(((float)L + 0.5f) * inv_nq);  // L / nq, exact for these sizes (L < 1024)
    int q = L - row * u.nq;
    const bool have = row < u.h;
    if (!have) row = 0, q = 0;
    BigItem o;
    o.voff = have ? (unsigned)((u.y0 + row) * W + u.x0 + 4 * q) * ES + (unsigned)cg * 4u * (unsigned)(H * W) * ES : kOffRange;
    o.meta = (unsigned)((row * u.pitch + 4 * q) * Big<1>::kSlotB + cg) | (have ? 1u << 17 : 0u);
    return o;
}

template <typename FT>
struct BigRegs {
    typename QuadRaw<FT>::type a[4];
};

template <typename FT>
__device__ __forceinline__ void big_issue(BigRegs<FT> &r, rsrc_t f1r, unsigned chunk_off, int H, int W, const BigItem &it) {
    const unsigned plane4 = (unsigned)(H * W) * (unsigned)sizeof(FT);
#pragma unroll
    for (int j = 0; j < 4; ++j) r.a[j] = QuadRaw<FT>::load(f1r, it.voff, chunk_off + (unsigned)j * plane4);
}

template <typename FT>
__device__ __forceinline__ void big_commit(float4 *s4, const BigRegs<FT> &r, const BigItem &it) {
    if ((it.meta >> 17) & 1u) {
        float4 *dst = s4 + (it.meta & 0x1FFFu);
        const f32x4 w0 = QuadRaw<FT>::widen(r.a[0]), w1 = QuadRaw<FT>::widen(r.a[1]), w2 = QuadRaw<FT>::widen(r.a[2]), w3 = QuadRaw<FT>::widen(r.a[3]);
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k * Big<1>::kSlotB] = make_float4(w0[k], w1[k], w2[k], w3[k]);
    }
}

// Two vertically adjacent 4 x 16-cell tiles (rows row0 .. row0 + 7) of direction b, region u (the union of their two regions).
template <int R, int NCH8, typename FT>
__device__ __forceinline__ void lean_big(const LcParams &p, unsigned char *smem, const RowPlan &u, int b, int row0, int col0, int tid, int lane,
                                         int wave) {
    using BG = Big<R>;
    constexpr int ROUNDS = 4, C = 8 * NCH8;
    constexpr int PW = BG::PW, P = BG::P, NP = BG::NP, D = BG::D, K = BG::K, NC = BG::NC, DS = BG::DS, TS = BG::TS, kSkew = BG::kSkew, CS = BG::CS;
    constexpr int kSB = BG::kSlotB;
    float4 *s4 = reinterpret_cast<float4 *>(smem);
    float *dbuf = reinterpret_cast<float *>(smem);
    int *cellX0 = reinterpret_cast<int *>(smem + BG::kStage);
    int *cellY0 = cellX0 + NC;
    float *cellNx = reinterpret_cast<float *>(cellY0 + NC);
    float *cellNy = cellNx + NC;
    int *cellFlag = reinterpret_cast<int *>(cellNy + NC);
    int *hdr = cellFlag + NC;  // [4], [5]: flagged cells of the two sub-tiles
    float *tab = reinterpret_cast<float *>(smem + BG::kStage + BG::kCellBytes);
    float *f0s = reinterpret_cast<float *>(smem + BG::kStage + BG::kCellBytes + BG::kTabBytes);

    const int G = p.G, H = p.H, W = p.W;
    const float xhi = p.win_xhi, xlo = -xhi, yhi = p.win_yhi, ylo = -yhi;
    const unsigned GG4 = (unsigned)(G * G) * 4u;

    // ---- flows (waves 0 and 1: one sub-tile each, lane = cell id inside it), the first f0 chunk and the first stage items go out at once
    const int sub = wave & 1;
    const int my_gi = row0 + 4 * sub + cell_row(lane), my_gj = col0 + cell_col(lane);
    const bool my_ok = (my_gi < G) & (my_gj < G);
    float my_nx = 0.f, my_ny = 0.f;
    if (wave < 2) {  // scalar
        const rsrc_t flr = make_rsrc(p.flow + (size_t)b * 2 * G * G, 2u * GG4);
        const unsigned fo = my_ok ? (unsigned)(my_gi * G + my_gj) * 4u : 0u;
        my_nx = buf_ld(flr, fo, 0u);
        my_ny = buf_ld(flr, fo, GG4);
    }
    // f0: wave w takes channel w of the chunk; lanes 0-31 = (sub-tile, tile row, column quad): one 16-byte load = four cells of a grid row
    const int fsub = lane >> 4, fr = (lane & 15) >> 2, fqd = lane & 3;
    const bool f0_lane = lane < 32;
    const rsrc_t f0r = make_rsrc(p.f0 + (size_t)b * p.f0_bs, (unsigned)C * GG4);
    auto f0_row = [](int cell) { return cell ^ ((cell >> 3) & 3); };  // (see lean_tile: spreads the filing over the banks)
    f32x4 f0q;
    auto f0_issue = [&](int c0) {
        const int gi = row0 + 4 * fsub + fr, gj = col0 + 4 * fqd;
        const bool in = f0_lane & (gi < G) & (gj < G);
        f0q = buf_ld4(f0r, in ? (unsigned)(gi * G + gj) * 4u + (unsigned)(c0 + wave) * GG4 : kOffRange, 0u);
    };
    auto f0_commit = [&]() {
        if (f0_lane) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int fc = 4 * fqd + e;
                const int cell = (fsub << 6) | ((fc >> 3) << 5) | (fr << 3) | (fc & 7);
                const bool fok = (row0 + 4 * fsub + fr < G) & (col0 + fc < G);
                f0s[f0_row(cell) * CS + wave] = fok ? f0q[e] : 0.f;
            }
        }
    };
    f0_issue(0);
    const int ipw = (u.nitems + 7) >> 3;  // items per wave (1 for the usual region)
    const BigItem it0 = big_item<FT>(u, H, W, wave, lane, 0);
    const rsrc_t f1r = make_rsrc(f1_of<FT>(p, b), (unsigned)C * (unsigned)(H * W) * (unsigned)sizeof(FT));
    BigRegs<FT> pre;
    big_issue<FT>(pre, f1r, 0u, H, W, it0);
    auto rest = [&](unsigned chunk_off) {  // regions of more than 256 quads: the further items one by one
        for (int k = 1; k < ipw; ++k) {
            const BigItem it = big_item<FT>(u, H, W, wave, lane, k);
            BigRegs<FT> r;
            big_issue<FT>(r, f1r, chunk_off, H, W, it);
            big_commit<FT>(s4, r, it);
        }
    };

    // ---- per-cell set-up (waves 0, 1) ---------------------------------------------------------------------------------------------
    if (wave < 2) {  // scalar
        const CellBox c = cell_box<PW>(my_ok, my_ok ? my_nx : 0.f, my_ok ? my_ny : 0.f, xlo, ylo, W, H);
        const int idx = (sub << 6) | lane;
        cellX0[idx] = c.X0;
        cellY0[idx] = c.Y0;
        cellNx[idx] = my_ok ? my_nx : 0.f;
        cellNy[idx] = my_ok ? my_ny : 0.f;
        cellFlag[idx] = c.flag;
        const unsigned long long slow_mask = __ballot(c.flag == kCellSlow);
        if (lane == 0) hdr[4 + sub] = __popcll(slow_mask);
    }
    f0_commit();
    big_commit<FT>(s4, pre, it0);
    rest(0u);
    __syncthreads();

    // ---- fraction table (lane = cell of a sub-tile, wave = tap index) ---------------------------------------------------------------
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const int cell = (s2 << 6) | lane;
        const float cnx = cellNx[cell], cny = cellNy[cell];
        const int cX0 = cellX0[cell], cY0 = cellY0[cell];
        bool tab_bad = false;
        constexpr int NTAB = (2 * D + kWaves - 1) / kWaves;
#pragma unroll
        for (int n = 0; n < NTAB; ++n) {
            const int a = wave + n * kWaves;  // scalar
            if (a < 2 * D) {
                const bool isy = a >= D;
                const int k = isy ? a - D : a;
                const float lin = isy ? gfn::linspace_step_at(ylo, yhi, p.win_ystep, D, k) : gfn::linspace_step_at(xlo, xhi, p.win_xstep, D, k);
                const float pix = unnorm((isy ? cny : cnx) + lin, isy ? H : W);
                const float fl = floorf(pix);
                const int origin = isy ? cY0 : cX0;
                tab_bad |= (origin != kFar) & !(fl == (float)(origin + k));
                tab[cell * TS + (cell >> 5) * kSkew + a] = pix - fl;
            }
        }
        if (tab_bad && atomicOr(&cellFlag[cell], kCellSlow) == 0) atomicAdd(&hdr[4 + s2], 1);  // rare
    }

    // ---- D-stage addressing: byte address of every (round, pass) patch position of the lane's cell ---------------------------------------
    int g, s16;
    lane_group(lane, g, s16);
    const int cr = wave * 4 + g;  // cell inside a 4 x 8 half (0..31); round rd: cell rd * 32 + cr
    unsigned ad[ROUNDS][NP];
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
        const int X0 = cellX0[rd * 32 + cr], Y0 = cellY0[rd * 32 + cr];
        const int base = X0 != kFar ? (Y0 - u.y0) * u.pitch + (X0 - u.x0) : 0;
#pragma unroll
        for (int t = 0; t < NP; ++t) {
            const int pp = s16 + 16 * t;
            const int yy = DivPW<PW>::div(pp), xx = pp - yy * PW;
            int slot = base + yy * u.pitch + xx;
            if (16 * t + 15 >= P) slot = pp < P ? slot : 0;
            ad[rd][t] = (unsigned)(slot * (kSB * 16));
        }
    }
    float acc[ROUNDS][NP];
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd)
#pragma unroll
        for (int t = 0; t < NP; ++t) acc[rd][t] = 0.f;

    // ---- main loop: 8 channels at a time ---------------------------------------------------------------------------------------------
#pragma unroll
    for (int st = 0; st < NCH8; ++st) {
        const bool more = st + 1 < NCH8;
        const unsigned next_off = (unsigned)((st + 1) * 8) * (unsigned)(H * W) * (unsigned)sizeof(FT);
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd)
#pragma unroll
            for (int t = 0; t < NP; ++t) asm volatile("" : "+v"(ad[rd][t]));  // keep the addresses: no recomputation per chunk
        if (more) {  // next chunk's loads: in flight across this D-stage
            big_issue<FT>(pre, f1r, next_off, H, W, it0);
            f0_issue((st + 1) * 8);
        }
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            const float4 *fq = reinterpret_cast<const float4 *>(f0s + f0_row(rd * 32 + cr) * CS);
            const float4 a0 = fq[0], a1 = fq[1];
#pragma unroll
            for (int t = 0; t < NP; ++t) {
                const float4 *q = reinterpret_cast<const float4 *>(smem + ad[rd][t]);
                const float4 v0 = q[0], v1 = q[1];
                float a = acc[rd][t];
                a = fmaf(a0.x, v0.x, a); a = fmaf(a0.y, v0.y, a); a = fmaf(a0.z, v0.z, a); a = fmaf(a0.w, v0.w, a);
                a = fmaf(a1.x, v1.x, a); a = fmaf(a1.y, v1.y, a); a = fmaf(a1.z, v1.z, a); a = fmaf(a1.w, v1.w, a);
                acc[rd][t] = a;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd)
#pragma unroll
            for (int t = 0; t < NP; ++t) asm volatile("" : "+v"(acc[rd][t]));  // pins the FMAs above this point
        if (more) {
            __syncthreads();  // everyone is done reading this chunk's pixels and f0 block
            big_commit<FT>(s4, pre, it0);
            rest(next_off);
            f0_commit();
            __syncthreads();
        }
    }

    // ---- epilogue: D -> LDS, bilinear combination, coalesced stores ----------------------------------------------------------------------
    __syncthreads();
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
        const int cell = rd * 32 + cr;
#pragma unroll
        for (int t = 0; t < NP; ++t) {
            const int pp = s16 + 16 * t;
            if (pp < P) dbuf[cell * DS + rd * kSkew + pp] = acc[rd][t];
        }
    }
    __syncthreads();
    const rsrc_t outr = make_rsrc(p.out + (size_t)b * p.out_bs, (unsigned)K * GG4);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        // lane -> cell so that a wave stores whole 64-byte grid-row segments: tile row lane >> 4, column lane & 15
        const int er = lane >> 4, ec = lane & 15;
        const int cell = (s2 << 6) | ((ec >> 3) << 5) | (er << 3) | (ec & 7);
        const int gi = row0 + 4 * s2 + er, gj = col0 + ec;
        const int flag = cellFlag[cell];
        if ((gi < G) & (gj < G) & !(flag & kCellSlow)) {
            const bool empty = (flag & kCellEmpty) != 0;
            const float *dc = dbuf + cell * DS + (cell >> 5) * kSkew;
            const float *tc = tab + cell * TS + (cell >> 5) * kSkew;
            const unsigned goff = (unsigned)(gi * G + gj) * 4u;
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
                    const float *d = dc + ky * PW;
                    float m[PW];
#pragma unroll
                    for (int x = 0; x < PW; ++x) m[x] = fmaf(d[PW + x], wy1s, d[x] * wy0s);
#pragma unroll
                    for (int kx = 0; kx < D; ++kx) {
                        const float val = fmaf(m[kx + 1], wx1[kx], m[kx] * wx0[kx]);
                        buf_st_nt(outr, goff, (unsigned)(ky * D + kx) * GG4, empty ? 0.f : val);
                    }
                }
            }
        }
    }

    // ---- flagged cells: general per-tap routine (about one cell in 10^4) ---------------------------------------------------------------
    const int nslow = __builtin_amdgcn_readfirstlane(hdr[4] + hdr[5]);
    if (nslow != 0) {  // block-uniform, rare
        __syncthreads();
        if (tid == 0) {
            int n = 0;
            for (int cell = 0; cell < NC; ++cell)
                if ((cellFlag[cell] & kCellSlow) && (row0 + 4 * (cell >> 6) + cell_row(cell & 63) < G) && (col0 + cell_col(cell & 63) < G)) cellX0[n++] = cell;
            hdr[4] = n;
            atomicAdd(p.todo + 4, n);  // informational (bench.py: flagged_cell_frac)
        }
        __syncthreads();
        const int total = hdr[4] * K;
        for (int e = tid; e < total; e += kThreads) {
            const int cell = cellX0[e / K], k = e % K;
            const int gi = row0 + 4 * (cell >> 6) + cell_row(cell & 63), gj = col0 + cell_col(cell & 63);
            p.out[(size_t)b * p.out_bs + ((size_t)k * G + gi) * G + gj] = tap_general<FT>(p, b, gi, gj, k / D, k % D, D, cellNx[cell], cellNy[cell]);
        }
    }
}

// grid: one workgroup per PAIR of vertically adjacent tiles (+ the second-launch workers in front, as in local_corr_tile2_kernel)
template <int R, int NCH, typename FT>
__global__ __launch_bounds__(kThreads, Lean<R>::kMinWaves) void local_corr_big_kernel(LcParams p) {
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
    const unsigned wbig = gfn::xcd_remap(blockIdx.x - kLeanWorkers, gridDim.x - kLeanWorkers);
    const unsigned tyb_n = (unsigned)(p.tiles_y + 1) >> 1, per_b = (unsigned)p.tiles_x * tyb_n;
    const unsigned b = wbig / per_b, rem = wbig - b * per_b;
    const unsigned tyb = rem / (unsigned)p.tiles_x, tx = rem - tyb * (unsigned)p.tiles_x;
    const unsigned wtop = b * (unsigned)(p.tiles_x * p.tiles_y) + 2u * tyb * (unsigned)p.tiles_x + tx;
    const bool has_bot = 2u * tyb + 1u < (unsigned)p.tiles_y;
    const unsigned wbot = has_bot ? wtop + (unsigned)p.tiles_x : wtop;
    typedef int i32x8 __attribute__((ext_vector_type(8)));
    typedef int i32x4s __attribute__((ext_vector_type(4)));
    i32x8 pt, pb;
    i32x4s gt;
    {
        const int *pp = p.plan + (size_t)wtop * kPlanInts, *pq = p.plan + (size_t)wbot * kPlanInts;
        asm volatile("s_load_dwordx8 %0, %3, 0x0\n\ts_load_dwordx4 %1, %3, 0x20\n\ts_load_dwordx8 %2, %4, 0x0\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(pt), "=&s"(gt), "=&s"(pb)
                     : "s"(pp), "s"(pq)
                     : "memory");
    }
    bool big = has_bot && !((pt[3] | pb[3]) & (kPlanSecond | kPlanHalves)) && ((pt[3] & pb[3] & kPlanInterior) != 0);
    RowPlan u;
    if (big) {
        const int ax0 = pt[0], ay0 = pt[1], aw = pt[2] & 0xffff, ah = pt[2] >> 16;
        const int bx0 = pb[0], by0 = pb[1], bw = pb[2] & 0xffff, bh = pb[2] >> 16;
        big = (aw > 0) & (ah > 0) & (bw > 0) & (bh > 0);
        u.x0 = min(ax0, bx0);
        u.y0 = min(ay0, by0);
        u.w = max(ax0 + aw, bx0 + bw) - u.x0;
        u.h = max(ay0 + ah, by0 + bh) - u.y0;
        big = big && big_region_fits<R>(u);
    }
    if (big) {
        lean_big<R, 2 * NCH, FT>(p, smem, u, gt[1], gt[2] & 0xffff, gt[2] >> 16, tid, lane, wave);
        return;
    }
    lean_small_tile<R, NCH, FT>(p, smem, wtop, tid, lane, wave);
    if (has_bot) {
        __syncthreads();  // the second tile reuses the LDS
        lean_small_tile<R, NCH, FT>(p, smem, wbot, tid, lane, wave);
    }
}
