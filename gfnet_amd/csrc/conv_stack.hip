// conv_stack.hip -- the refiners' conv stack on gfx950 ("next" after the hot path).
//
// Reference: ConvRefiner.create_block / forward, model/network.py:471-487 and :560-563 -- per block
//   Conv2d(C, C, 5x5, padding 2, groups=C)  ->  BatchNorm2d (eval)  ->  ReLU  ->  Conv2d(C, C, 1x1)
// nine of them (block1 + 8 hidden blocks) and a final Conv2d(C, 3, 1x1), on (2c+disp+K)-channel grid
// maps: C = 417/361/177/73/24 at scales 16/8/4/2/1.  On MI355X the PyTorch/MIOpen stack takes
// 147 ms (fp16 autocast) / 180 ms (fp32) per 32-pair step -- 15x the whole correlation/sampling/solve
// path -- mostly in the small-C, large-grid scales where the depthwise convs are launch/latency bound.
//
// One conv block = one kernel (dwpw_fused_kernel): a workgroup owns a 128-cell tile of one map and a
// slab of output channels.  Per tile of 16 input channels it stages the cells' halo (zero padded) in
// LDS, computes t = relu((dw5x5(x) + conv_bias) * alpha + beta) on the VALU (alpha/beta = eval-mode
// BatchNorm folded by the packer) straight into the B-operand tile of the 1x1 conv, and runs
// y += W[:, k-tile] . t on the fp32 matrix core (v_mfma_f32_32x32x2_f32: exact fp32 products and
// accumulation, the numerics class of the reference's fp32 CPU path).  The intermediate t never
// leaves the CU: HBM traffic is one read of x and one write of y per block.  The next channel tile's
// halo and weight tile are prefetched into registers under the current tile's arithmetic.
//
// A two-pass variant (dw5x5 kernel -> t in HBM -> GEMM kernel) computes bit-identical results; it
// serves grids whose side is not a multiple of 4 and is the ablation/parity partner of the fused one.
#include "common.h"
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kKT = 16;    // channels per K tile
constexpr int kNP = 8;     // channel pairs per K tile
constexpr int kBN = 128;   // cells per workgroup tile: 4 waves x 32
constexpr int kCP2 = 64;   // floats per channel PAIR in the packed depthwise parameters

__host__ __device__ inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// ---- packed parameters of one block ---------------------------------------------------------------
// [ cp: Kp/2 x 64 ][ wt: Kp x Mp weights, zero padded, in MFMA operand order ][ bias: Mp ],  Kp = ceil16(C), Mp = ceil32(M)
// wt: W[m][k] sits at wt_index(k, m): per K tile of 16 the order is [k half sg][k parity kh][m][j], k = 16*kt + 8*sg + 2*j + kh,
// so that one 16-byte LDS read hands a lane its A operands of four consecutive MFMA k-steps.
// cp row of channel pair (2p, 2p+1): float 2t+h = parameter t of channel 2p+h; t = 0..24 the 5x5 taps
// (row major), 25 = conv bias, 26 = alpha, 27 = beta -- the layout the packed-fp32 (v_pk_fma_f32)
// depthwise loop reads as register pairs.
struct PackDims {
    int Kp, Mp;
    __host__ __device__ PackDims(int C, int M) : Kp(round_up(C, kKT)), Mp(round_up(M, 32)) {}
    __host__ __device__ size_t cp_off() const { return 0; }
    __host__ __device__ size_t wt_off() const { return (size_t)(Kp / 2) * kCP2; }
    __host__ __device__ size_t bias_off() const { return wt_off() + (size_t)Kp * Mp; }
    __host__ __device__ size_t wt16_off() const { return bias_off() + Mp; }  // the same weights in fp16 (two per float slot)
    __host__ __device__ size_t total() const { return wt16_off() + (size_t)Kp * Mp / 2; }
    // fp16 weights, in halfs from wt16_off: per K tile [k half kg][m][8], k = 16*kt + 8*kg + j: one 16-byte read = a lane's
    // A operand of v_mfma_f32_32x32x16_f16
    __host__ __device__ size_t wt16_index(int k, int m) const { return ((size_t)((k >> 4) * 2 + ((k >> 3) & 1)) * Mp + m) * 8 + (k & 7); }
    __host__ __device__ size_t wt_index(int k, int m) const {
        const int kt = k >> 4, r = k & 15, sg = r >> 3, kh = r & 1, j = (r & 7) >> 1;
        return ((size_t)((kt * 2 + sg) * 2 + kh) * Mp + m) * 4 + j;
    }
};

__global__ __launch_bounds__(256) void pack_block_kernel(const float *__restrict__ dw_w, const float *__restrict__ dw_b,
                                                         const float *__restrict__ alpha, const float *__restrict__ beta,
                                                         const float *__restrict__ pw_w, const float *__restrict__ pw_b,
                                                         float *__restrict__ packed, int C, int M) {
    const PackDims d(C, M);
    const size_t total = d.total();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (i < d.wt_off()) {
            const int pair = (int)(i / kCP2), j = (int)(i % kCP2);
            const int t = j >> 1, k = 2 * pair + (j & 1);
            if (k < C) {
                if (t < 25) v = dw_w[(size_t)k * 25 + t];
                else if (t == 25) v = dw_b ? dw_b[k] : 0.f;
                else if (t == 26) v = alpha[k];
                else if (t == 27) v = beta[k];
            }
        } else if (i < d.bias_off()) {
            const size_t e = i - d.wt_off();  // invert wt_index
            const int j = (int)(e & 3), m = (int)((e >> 2) % d.Mp), g = (int)((e >> 2) / d.Mp);
            const int k = (g >> 2) * 16 + ((g >> 1) & 1) * 8 + 2 * j + (g & 1);
            if (k < C && m < M) v = pw_w[(size_t)m * C + k];
        } else if (i < d.wt16_off()) {
            const int m = (int)(i - d.bias_off());
            if (m < M) v = pw_b[m];
        } else {  // invert wt16_index for the two halfs of this slot
            _Float16 h[2];
            for (int e2 = 0; e2 < 2; ++e2) {
                const size_t hidx = 2 * (i - d.wt16_off()) + e2;
                const int j = (int)(hidx & 7), m = (int)((hidx >> 3) % d.Mp), g = (int)((hidx >> 3) / d.Mp);
                const int k = (g >> 1) * 16 + (g & 1) * 8 + j;
                h[e2] = (k < C && m < M) ? (_Float16)pw_w[(size_t)m * C + k] : (_Float16)0.f;
            }
            __builtin_memcpy(&v, h, 4);
        }
        packed[i] = v;
    }
}

// The depthwise arithmetic, shared by both variants so that they agree bit for bit: 25 fmas per
// output in (dy, dx) order, then (acc + bias) * alpha + beta, relu.
__device__ __forceinline__ float dw_finish(float acc, float cb, float al, float be) { return fmaxf((acc + cb) * al + be, 0.f); }

__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 dw_finish2(f32x2 acc, f32x2 cb, f32x2 al, f32x2 be) {
    f32x2 r;
    r.x = dw_finish(acc.x, cb.x, al.x, be.x);
    r.y = dw_finish(acc.y, cb.y, al.y, be.y);
    return r;
}

// ---- fused block ------------------------------------------------------------------------------------
// Workgroup = NS*4 waves on one 128-cell tile of one map and NS slabs of 32*MT output channels (the
// waves of both slabs share the depthwise work and the B operand tile).  Per K tile of 16 channels:
//   commit   registers -> LDS: the tile's halo, channel-pair interleaved ([pair][row][cell][2]), the weight
//            tile W^T[k][m], the pairs' depthwise parameters        | barrier
//   issue    global loads of the NEXT K tile into registers (land under the arithmetic below)
//   depthwise  per thread 4/NS cells x one channel pair on packed fp32 (v_pk_fma_f32), relu, into the
//            B operand tile Bs[channel][cell]                        | barrier
//   matrix   8 k-steps x MT v_mfma_f32_32x32x2_f32 per wave
// Halo cells outside the map are zeroed once in LDS and never written (zero padding for free).
template <int MT, int TW, int NS, bool F16, int NB>
__global__ __launch_bounds__(256 * NS, NS == 1 ? 2 : 1) void dwpw_fused_kernel(const float *__restrict__ x,
                                                                               const float *__restrict__ packed,
                                                                               float *__restrict__ y, int M, int K, int G,
                                                                               int tiles_x, int tiles_y, int ngrp, unsigned nwork,
                                                                               int tpb, int dbg_arg) {
#ifdef GFN_ABLATE  // timing experiments only (tools/ablate_convblock.py): skip parts of the kernel; results are wrong
    const int dbg = dbg_arg;
#else
    constexpr int dbg = 0;
#endif
    static_assert(NB == 1 || (NB == 2 && NS == 1 && MT <= 3), "256-cell tiles: one slab of at most 96 output channels");
    constexpr int NT = 256 * NS;
    constexpr int BN = kBN * NB;            // cells per workgroup tile
    constexpr int TH = BN / TW;             // tile rows
    constexpr int HR = TH + 4;              // halo rows
    constexpr int RV4 = (TW + 8) / 4;       // float4 per staged halo row and channel: cells col0-4 .. col0+TW+3
    constexpr int RPP = (TW + 8) * 2 + 4;   // LDS floats per halo row of a channel pair (+4: bank spread)
    constexpr int PP = HR * RPP;            // LDS floats per channel pair
    constexpr int PS = kNP * HR * RV4;      // staging slots: one = the same 4 cells of both channels of a pair
    constexpr int XPP = (PS + NT - 1) / NT;
    constexpr int BM = 32 * MT, BMS = BM * NS;
    constexpr int GP = F16 ? 2 : 4;         // 16-byte operand groups per weight row and K tile
    constexpr int AV4 = GP * BMS;           // 16-byte pieces of one weight tile
    constexpr int APT = (AV4 + NT - 1) / NT;
    constexpr int PPT = kNP * kCP2 / NT;    // parameter floats per thread and K tile
    constexpr int CPT = 4 / NS;             // depthwise: cells per thread and row
    constexpr int RB = NB;                  //            rows per thread
    constexpr int TPP = kBN / CPT;          //            threads per channel pair
    constexpr int GPR = TW / CPT;           //            threads per tile row
    // Two-row depthwise threads step through the halo two rows at a time, and two row pitches are 0 mod 8 banks:
    // every other row PAIR is stored 16 bytes later (the pitch has the room), which puts the 16 lanes of one
    // ds_read_b128 group back on all 32 banks (measured: 69% of LDS cycles were bank conflicts without it).
    constexpr bool SWZ = NB == 2;

    __shared__ __attribute__((aligned(16))) float Xs[kNP * PP];
    // fp32: [buf][(sg*2+kh)*BMS + m] = A operands of k-steps 4sg .. 4sg+3;  fp16: [buf][kg*BMS + m] = 8 halfs k = 8kg ..
    __shared__ float4 As4[2][GP * BMS];
    // B operand tile: fp32 [channel k][cell]; fp16 [pair][cell] of half2 (channels 2p, 2p+1) -- consecutive depthwise
    // threads write consecutive 16-byte pieces, the matrix lanes read consecutive dwords
    __shared__ __attribute__((aligned(16))) float Bs[F16 ? kNP * BN : kKT * BN];
    __shared__ __attribute__((aligned(16))) float Ps[kNP * kCP2];

    const PackDims pd(K, M);
    const float *cp = packed + pd.cp_off();
    const float *wt = packed + pd.wt_off();
    const float *bias = packed + pd.bias_off();
    const int Mp = pd.Mp, Kp = pd.Kp;
    const int plane = G * G;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // a workgroup walks `tpb` consecutive work items (cell tile x slab group); the (item, K tile) pairs form one
    // pipeline, so only the first item's load latency is exposed
    const unsigned lb = gfn::xcd_remap(blockIdx.x, gridDim.x);
    const unsigned w_begin = lb * (unsigned)tpb;
    const unsigned w_end = w_begin + (unsigned)tpb < nwork ? w_begin + (unsigned)tpb : nwork;
    const int nk = Kp / kKT;
    const int total = (int)(w_end - w_begin) * nk;
    auto decode = [&](unsigned item, int &b, int &row0, int &col0, int &m0) {
        const unsigned grp = item % (unsigned)ngrp;
        item /= (unsigned)ngrp;
        const unsigned tx = item % (unsigned)tiles_x;
        item /= (unsigned)tiles_x;
        const unsigned ty = item % (unsigned)tiles_y;
        b = (int)(item / (unsigned)tiles_y);
        row0 = (int)ty * TH, col0 = (int)tx * TW, m0 = (int)grp * BMS;
    };

    // staging slots of this thread: LDS side fixed, global side per item
    int xl[XPP], xp2[XPP], xg[XPP];
#pragma unroll
    for (int i = 0; i < XPP; ++i) {
        const int e = tid + NT * i;
        const int p = e / (HR * RV4), rem = e - p * (HR * RV4);
        const int hr = rem / RV4, q = rem - hr * RV4;
        xp2[i] = 2 * p;
        xl[i] = p * PP + hr * RPP + 8 * q + (SWZ && ((hr >> 1) & 1) ? 4 : 0);
    }
    unsigned l_item = w_begin;  // load stage: the (item, K tile) the next issue() fetches
    int l_kt = 0, l_m0 = 0, l_valid = 0;  // l_valid bit i: slot i lies inside the map (else zero padding)
    const float *l_xb = x;
    auto load_stage_enter_item = [&]() {
        int b, row0, col0;
        decode(l_item, b, row0, col0, l_m0);
        l_xb = x + (size_t)b * K * plane;
        l_valid = 0;
#pragma unroll
        for (int i = 0; i < XPP; ++i) {
            const int e = tid + NT * i;
            const int p = e / (HR * RV4), rem = e - p * (HR * RV4);
            const int hr = rem / RV4, q = rem - hr * RV4;
            const int gy = row0 - 2 + hr, gx = col0 - 4 + 4 * q;
            const bool ok = e < PS && (unsigned)gy < (unsigned)G && gx >= 0 && gx < G;  // G % 4 == 0: 4 cells in or out together
            xg[i] = ok ? gy * G + gx : 0;
            l_valid |= ok ? 1 << i : 0;
        }
    };
    if (!(dbg & 32))
        for (int e = tid; e < kNP * PP / 4; e += NT) reinterpret_cast<float4 *>(Xs)[e] = make_float4(0.f, 0.f, 0.f, 0.f);

    static_assert(APT <= 4, "weight tile slots");
    float4 xr0[XPP], xr1[XPP];
    float4 ar0, ar1, ar2, ar3;  // named, not an array: the compiler demotes a float4 array here to LDS
    float pr[PPT];
    int r_valid = 0;  // l_valid of the item in the registers; bit 8: its first K tile (refresh the zero padding)
    const float4 *wt4 = reinterpret_cast<const float4 *>(F16 ? packed + pd.wt16_off() : wt);
    auto a_load = [&](int kt, int i) {
        const int e = tid + NT * i;
        const int g = e / BMS, m = l_m0 + e - g * BMS;
        const bool ok = e < AV4 && m < Mp;
        return wt4[(size_t)(kt * GP + (ok ? g : 0)) * Mp + (ok ? m : 0)];
    };
    auto a_store = [&](int buf, int i, const float4 &v) {
        const int e = tid + NT * i;
        if (e < AV4) As4[buf][e] = v;
    };
    auto issue = [&]() {  // fetch (l_item, l_kt) into registers, advance the load stage
        if (dbg & 1) return;
        const int k0 = l_kt * kKT;
#pragma unroll
        for (int i = 0; i < XPP; ++i) {
            const int c0 = min(k0 + xp2[i], K - 1), c1 = min(k0 + xp2[i] + 1, K - 1);  // past C: any finite data, its taps are 0
            xr0[i] = *reinterpret_cast<const float4 *>(l_xb + c0 * plane + xg[i]);
            xr1[i] = *reinterpret_cast<const float4 *>(l_xb + c1 * plane + xg[i]);
        }
        if constexpr (APT > 0) ar0 = a_load(l_kt, 0);
        if constexpr (APT > 1) ar1 = a_load(l_kt, 1);
        if constexpr (APT > 2) ar2 = a_load(l_kt, 2);
        if constexpr (APT > 3) ar3 = a_load(l_kt, 3);
#pragma unroll
        for (int i = 0; i < PPT; ++i) pr[i] = cp[(size_t)(k0 / 2) * kCP2 + tid + NT * i];
        r_valid = l_valid | (l_kt == 0 && l_item != w_begin ? 256 : 0);
        if (++l_kt == nk) {
            l_kt = 0;
            if (++l_item < w_end) load_stage_enter_item();
        }
    };
    auto commit = [&](int buf) {
        if (dbg & 16) return;
#pragma unroll
        for (int i = 0; i < XPP; ++i) {
            if (r_valid & (1 << i)) {
                const f32x4 lo = {xr0[i].x, xr1[i].x, xr0[i].y, xr1[i].y}, hi = {xr0[i].z, xr1[i].z, xr0[i].w, xr1[i].w};
                *reinterpret_cast<f32x4 *>(&Xs[xl[i]]) = lo;
                *reinterpret_cast<f32x4 *>(&Xs[xl[i] + 4]) = hi;
            } else if ((r_valid & 256) && tid + NT * i < PS) {  // outside the map: re-zero once per item after the first
                *reinterpret_cast<float4 *>(&Xs[xl[i]]) = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4 *>(&Xs[xl[i] + 4]) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if constexpr (APT > 0) a_store(buf, 0, ar0);
        if constexpr (APT > 1) a_store(buf, 1, ar1);
        if constexpr (APT > 2) a_store(buf, 2, ar2);
        if constexpr (APT > 3) a_store(buf, 3, ar3);
#pragma unroll
        for (int i = 0; i < PPT; ++i) Ps[tid + NT * i] = pr[i];
    };

    f32x16 acc[NB * MT];  // [cell half g][row tile i]
#pragma unroll
    for (int i = 0; i < NB * MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int col = lane & 31, kh = lane >> 5;
    const int slab = wave >> 2, cw = wave & 3;  // matrix role: output slab, 32-cell group (of each 128-cell half)
    // depthwise role: channel pair dp, tile rows dr .. dr+RB-1, cells dc .. dc+CPT-1
    // The LDS serves a ds_read_b128 in 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32): with 32-wide tiles the
    // lanes of one group are given one row (2-cell threads) or two rows (4-cell threads) of a channel pair, so that their
    // reads are consecutive 16-byte units (measured 4.8 vs 8.5 LDS cycles per read instruction with lanes in natural order)
    const int l32 = lane & 31;
    const int lperm = TW == 32 ? (int)((0x73261540u >> (4 * (l32 >> 2))) & 7u) * 4 + (l32 & 3) + (lane & 32) : lane;
    const int dtid = (tid & ~63) | lperm;
    const int dp = dtid / TPP, dg = dtid - dp * TPP, dr = (dg / GPR) * RB, dc = (dg - (dg / GPR) * GPR) * CPT;
    // staged cell index = tile cell + 4; taps reach cells dc-2 ..; dw_src: halo rows hy with (hy>>1) even, dw_src1: odd (see SWZ)
    const int dw_shift = SWZ ? 4 * ((dr >> 1) & 1) : 0;
    const float *dw_src = &Xs[dp * PP + dr * RPP + 2 * (dc + 2) + dw_shift];
    const float *dw_src1 = &Xs[dp * PP + dr * RPP + 2 * (dc + 2) + (SWZ ? 4 - dw_shift : 0)];
    const f32x2 *dw_par = reinterpret_cast<const f32x2 *>(&Ps[dp * kCP2]);
    float *dw_dst = &Bs[(F16 ? dp : 2 * dp) * BN + dr * TW + dc];  // fp32: channel 2dp here, 2dp+1 one plane (BN) further

    if (total <= 0) return;
    load_stage_enter_item();
    issue();
    __syncthreads();  // Xs zeroed
    unsigned c_item = w_begin;  // the item being accumulated
    int c_kt = 0;
    for (int t = 0; t < total; ++t) {
        const int buf = t & 1;
        commit(buf);
        __syncthreads();
        if (t + 1 < total) issue();
        if (!(dbg & 2)) {  // depthwise: RB output rows x CPT cells x one channel pair; every halo row and every tap row is read once
            f32x2 a[RB][CPT];
#pragma unroll
            for (int ro = 0; ro < RB; ++ro)
#pragma unroll
                for (int j = 0; j < CPT; ++j) a[ro][j] = f32x2{0.f, 0.f};
            f32x2 w[2][5];  // tap rows hy and hy-1
#pragma unroll
            for (int hy = 0; hy < 4 + RB; ++hy) {
                f32x2 v[CPT + 4];
#pragma unroll
                for (int q = 0; q < (CPT + 4) / 2; ++q) {
                    const float4 f = *reinterpret_cast<const float4 *>((((hy >> 1) & 1) ? dw_src1 : dw_src) + hy * RPP + 4 * q);
                    v[2 * q] = f32x2{f.x, f.y};
                    v[2 * q + 1] = f32x2{f.z, f.w};
                }
                if (hy < 5) {
#pragma unroll
                    for (int dx = 0; dx < 5; ++dx) w[hy & 1][dx] = dw_par[hy * 5 + dx];
                }
#pragma unroll
                for (int ro = 0; ro < RB; ++ro) {
                    const int dy = hy - ro;  // tap row that maps halo row hy onto output row ro
                    if (dy < 0 || dy > 4) continue;
#pragma unroll
                    for (int dx = 0; dx < 5; ++dx)
#pragma unroll
                        for (int j = 0; j < CPT; ++j) a[ro][j] = pk_fma(w[dy & 1][dx], v[j + dx], a[ro][j]);
                }
            }
            const f32x2 cb = dw_par[25], al = dw_par[26], be = dw_par[27];
#pragma unroll
            for (int ro = 0; ro < RB; ++ro) {
                f32x2 t[CPT];
#pragma unroll
                for (int j = 0; j < CPT; ++j) t[j] = dw_finish2(a[ro][j], cb, al, be);
                float *dst = dw_dst + ro * TW;
                if constexpr (F16) {  // CPT cells x half2: one 16-byte (8-byte) write
                    if constexpr (CPT == 4) {
                        const f16x8 h = {(_Float16)t[0].x, (_Float16)t[0].y, (_Float16)t[1].x, (_Float16)t[1].y,
                                         (_Float16)t[2].x, (_Float16)t[2].y, (_Float16)t[3].x, (_Float16)t[3].y};
                        *reinterpret_cast<f32x4 *>(dst) = __builtin_bit_cast(f32x4, h);
                    } else {
                        const f16x4 h = {(_Float16)t[0].x, (_Float16)t[0].y, (_Float16)t[1].x, (_Float16)t[1].y};
                        *reinterpret_cast<f32x2 *>(dst) = __builtin_bit_cast(f32x2, h);
                    }
                } else if constexpr (CPT == 4) {
                    *reinterpret_cast<float4 *>(dst) = make_float4(t[0].x, t[1].x, t[2].x, t[3].x);
                    *reinterpret_cast<float4 *>(dst + BN) = make_float4(t[0].y, t[1].y, t[2].y, t[3].y);
                } else {
                    *reinterpret_cast<float2 *>(dst) = make_float2(t[0].x, t[1].x);
                    *reinterpret_cast<float2 *>(dst + BN) = make_float2(t[0].y, t[1].y);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < NB; ++g) {
            if (dbg & 4) break;
            const int cell = (g * 4 + cw) * 32 + col;  // this lane's B column
            if constexpr (F16) {  // one v_mfma_f32_32x32x16_f16 per row tile: lane (n or m = lane&31, kg = lane>>5) holds 8 halfs
                // channel pairs 4kh .. 4kh+3 of this cell = halfs k = 8kh .. 8kh+7
                const float *bp = &Bs[4 * kh * BN + cell];
                const f32x4 bq = {bp[0], bp[BN], bp[2 * BN], bp[3 * BN]};
                const f16x8 bv = __builtin_bit_cast(f16x8, bq);
                const f16x8 *asrc = reinterpret_cast<const f16x8 *>(&As4[buf][kh * BMS + slab * BM + col]);
                f16x8 av[MT];
#pragma unroll
                for (int i = 0; i < MT; ++i) av[i] = asrc[i * 32];
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[g * MT + i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[i], bv, acc[g * MT + i], 0, 0, 0);
            } else {
                // B operands of all 8 k-steps and the A operands of k-steps 0..3 are fetched up front; each row
                // tile's A operands of k-steps 4..7 are fetched as soon as its first four MFMAs are issued
                const float *bsrc = &Bs[kh * BN + cell];  // t[2s+kh][cell] at + s*2*BN
                const float4 *asrc = &As4[buf][kh * BMS + slab * BM + col];
                float bv[8];
                float4 av[MT];
#pragma unroll
                for (int s8 = 0; s8 < 8; ++s8) bv[s8] = bsrc[s8 * 2 * BN];
#pragma unroll
                for (int i = 0; i < MT; ++i) av[i] = asrc[i * 32];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    f32x16 &c = acc[g * MT + i];
                    c = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, bv[0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, bv[1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, bv[2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, bv[3], c, 0, 0, 0);
                    av[i] = asrc[2 * BMS + i * 32];
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    f32x16 &c = acc[g * MT + i];
                    c = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, bv[4], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, bv[5], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, bv[6], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, bv[7], c, 0, 0, 0);
                }
            }
        }
        if (++c_kt < nk) continue;
        // item finished: D[row][col], col = lane&31 -> cell (g*4+cw)*32+col of the tile, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
        int b, row0, col0, m0;
        decode(c_item, b, row0, col0, m0);
        c_kt = 0;
        ++c_item;
#pragma unroll
    for (int g = 0; g < NB; ++g) {
        const int p = (g * 4 + cw) * 32 + col;
        const int gy = row0 + p / TW, gx = col0 + p % TW;
        if (gy >= G || gx >= G || ((dbg & 8) && acc[0][0] != 12345.f)) continue;
        float *yb = y + (size_t)b * M * plane + (size_t)gy * G + gx;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int mb = m0 + slab * BM + i * 32;  // 32 output channels of this accumulator tile
            if (mb >= M) break;
            float4 bq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const float4 *>(bias + mb + 8 * q + 4 * kh);  // bias is padded to Mp
            float *yt = yb + (size_t)(mb + 4 * kh) * plane;
            if (mb + 32 <= M) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    yt[(size_t)(8 * q) * plane] = acc[g * MT + i][4 * q] + bq[q].x;
                    yt[(size_t)(8 * q + 1) * plane] = acc[g * MT + i][4 * q + 1] + bq[q].y;
                    yt[(size_t)(8 * q + 2) * plane] = acc[g * MT + i][4 * q + 2] + bq[q].z;
                    yt[(size_t)(8 * q + 3) * plane] = acc[g * MT + i][4 * q + 3] + bq[q].w;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = mb + 8 * q + 4 * kh;
                    if (m < M) yt[(size_t)(8 * q) * plane] = acc[g * MT + i][4 * q] + bq[q].x;
                    if (m + 1 < M) yt[(size_t)(8 * q + 1) * plane] = acc[g * MT + i][4 * q + 1] + bq[q].y;
                    if (m + 2 < M) yt[(size_t)(8 * q + 2) * plane] = acc[g * MT + i][4 * q + 2] + bq[q].z;
                    if (m + 3 < M) yt[(size_t)(8 * q + 3) * plane] = acc[g * MT + i][4 * q + 3] + bq[q].w;
                }
            }
        }
    }
#pragma unroll
        for (int i = 0; i < NB * MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    }
}

// ---- two-pass variant -------------------------------------------------------------------------------
// depthwise 5x5 + affine + relu; one workgroup works inside one (b, c) plane so the 28 parameters are
// wave-uniform.  VEC = 4: one thread = 4 cells of a row (G % 4 == 0); VEC = 1: any G.
template <int VEC>
__global__ __launch_bounds__(256) void dw5x5_kernel(const float *__restrict__ x, float *__restrict__ t,
                                                    const float *__restrict__ packed, int C, int G, int bpp) {
    const int pl = blockIdx.x / bpp;
    const int c = pl % C;
    const int local = (blockIdx.x - pl * bpp) * 256 + threadIdx.x;
    const int GV = G / VEC;
    if (local >= G * GV) return;
    const int i = local / GV, jv = local - i * GV;
    const float *xp = x + (size_t)pl * G * G;
    const float *wc = packed + (size_t)(c >> 1) * kCP2 + (c & 1);  // parameter t of this channel at wc[2t]
    float acc[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[q] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 5; ++dy) {
        const int yy = i + dy - 2;
        const bool rok = (unsigned)yy < (unsigned)G;
        const float *row = xp + (size_t)(rok ? yy : 0) * G;
        if constexpr (VEC == 4) {
            float v[12];
            const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 *r4 = reinterpret_cast<const float4 *>(row);
            const bool lok = rok && jv > 0, rok2 = rok && jv + 1 < GV;
            float4 a = r4[lok ? jv - 1 : 0], m = r4[jv], e = r4[rok2 ? jv + 1 : 0];
            a = lok ? a : zero, m = rok ? m : zero, e = rok2 ? e : zero;
            v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = m.x, v[5] = m.y, v[6] = m.z, v[7] = m.w;
            v[8] = e.x, v[9] = e.y, v[10] = e.z, v[11] = e.w;  // cells 4jv-4 .. 4jv+7
#pragma unroll
            for (int dx = 0; dx < 5; ++dx)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = fmaf(wc[2 * (dy * 5 + dx)], v[q + dx + 2], acc[q]);
        } else {
#pragma unroll
            for (int dx = 0; dx < 5; ++dx) {
                const int xx = jv + dx - 2;
                const bool ok = rok && (unsigned)xx < (unsigned)G;
                const float xv = row[ok ? xx : 0];
                acc[0] = fmaf(wc[2 * (dy * 5 + dx)], ok ? xv : 0.f, acc[0]);
            }
        }
    }
    float *dst = t + (size_t)pl * G * G + (size_t)i * G + (size_t)jv * VEC;
#pragma unroll
    for (int q = 0; q < VEC; ++q) dst[q] = dw_finish(acc[q], wc[50], wc[52], wc[54]);
}

// y[b] = W . t[b] + bias on the matrix core; same operands, instruction and k order as the fused kernel.
template <int MT, bool F16>
__global__ __launch_bounds__(256, 2) void pw_gemm_kernel(const float *__restrict__ packed, const float *__restrict__ t,
                                                         float *__restrict__ y, int M, int K, int N) {
    constexpr int BM = 32 * MT;
    __shared__ __attribute__((aligned(16))) float As[kKT][BM];
    __shared__ __attribute__((aligned(16))) float Bs[kKT][kBN];
    const PackDims pd(K, M);
    const float *wt = packed + pd.wt_off();
    const float *bias = packed + pd.bias_off();
    const int Mp = pd.Mp, Kp = pd.Kp;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, m0 = blockIdx.y * BM, n0 = blockIdx.x * kBN;
    const float *tb = t + (size_t)b * K * N;
    const int col = lane & 31, kh = lane >> 5;
    f32x16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int k0 = 0; k0 < Kp; k0 += kKT) {
        for (int e = tid; e < kKT * BM; e += 256) {
            const int k = e / BM, mm = e - k * BM;
            As[k][mm] = m0 + mm < Mp ? wt[pd.wt_index(k0 + k, m0 + mm)] : 0.f;
        }
        for (int e = tid; e < kKT * (kBN / 4); e += 256) {
            const int k = e / (kBN / 4), n4 = e - k * (kBN / 4);
            const int kk = k0 + k, n = n0 + n4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kk < K) {
                const float *src = tb + (size_t)kk * N + n;
                if ((N & 3) == 0) {
                    if (n < N) v = *reinterpret_cast<const float4 *>(src);
                } else {
                    if (n < N) v.x = src[0];
                    if (n + 1 < N) v.y = src[1];
                    if (n + 2 < N) v.z = src[2];
                    if (n + 3 < N) v.w = src[3];
                }
            }
            *reinterpret_cast<float4 *>(&Bs[k][n4 * 4]) = v;
        }
        __syncthreads();
        if constexpr (F16) {
            const _Float16 *w16 = reinterpret_cast<const _Float16 *>(packed + pd.wt16_off());
            f16x8 bv;
#pragma unroll
            for (int j = 0; j < 8; ++j) bv[j] = (_Float16)Bs[8 * kh + j][wave * 32 + col];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int m = m0 + i * 32 + col;
                f16x8 av;
#pragma unroll
                for (int j = 0; j < 8; ++j) av[j] = m < Mp ? w16[pd.wt16_index(k0 + 8 * kh + j, m)] : (_Float16)0.f;
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[i], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < kKT / 2; ++s) {
                const float bv = Bs[2 * s + kh][wave * 32 + col];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const float av = As[2 * s + kh][i * 32 + col];
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    const int n = n0 + wave * 32 + col;
    if (n < N) {
        float *yb = y + (size_t)b * M * N + n;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (m < M) yb[(size_t)m * N] = acc[i][r] + bias[m];
            }
    }
}

// 1x1 conv with a handful of output channels (out_conv: C -> 3): one thread per cell, coalesced along
// the map for every input channel; output channels in groups of 4.
__global__ __launch_bounds__(256) void pw_small_kernel(const float *__restrict__ w, const float *__restrict__ t,
                                                       const float *__restrict__ bias, float *__restrict__ y, int B, int M,
                                                       int K, int N) {
    const long total = (long)B * N;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int n = (int)(idx % N);
        const int b = (int)(idx / N);
        const float *tp = t + (size_t)b * K * N + n;
        for (int mb = 0; mb < M; mb += 4) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < K; ++k) {
                const float v = tp[(size_t)k * N];
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    if (mb + m < M) acc[m] = fmaf(w[(size_t)(mb + m) * K + k], v, acc[m]);
            }
            for (int m = 0; m < 4 && mb + m < M; ++m) y[((size_t)b * M + mb + m) * N + n] = acc[m] + bias[mb + m];
        }
    }
}

inline unsigned grid_for(long total, int cap = 65536) {
    long g = (total + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

// output-channel slabs: as few workgroups along M as possible with <= 7 MFMA row tiles (112 accumulators)
inline void slab_shape(int M, int *nblk, int *mt) {
    const int tiles = (M + 31) / 32;
    *nblk = (tiles + 6) / 7;
    *mt = (tiles + *nblk - 1) / *nblk;
}

// GFN_CONV_TPB (environment, experiments): work items per workgroup of the fused kernel, 0 = heuristic
static int g_conv_tpb = [] {
    const char *e = getenv("GFN_CONV_TPB");
    return e ? atoi(e) : 0;
}();

template <int MT, int TW, int NS, bool F16, int NB>
int launch_fused_mt(const float *x, const float *packed, float *y, int B, int M, int K, int G, int dbg, hipStream_t s) {
    constexpr int TH = kBN * NB / TW;
    const int tiles_x = (G + TW - 1) / TW, tiles_y = (G + TH - 1) / TH;
    const int ngrp = (M + 32 * MT * NS - 1) / (32 * MT * NS);
    const long nwork = (long)B * tiles_x * tiles_y * ngrp;
    if (nwork > 0x7fffffffL) return gfn::fail(GFN_ERR_INVALID_ARG, "conv_block: too many tiles");
    // work items per workgroup (pipelined back to back)
    int tpb = nwork >= 16384 ? 2 : 1;  // measured: pays only on the largest grids
    if (g_conv_tpb > 0) tpb = g_conv_tpb;
    const unsigned grid = (unsigned)((nwork + tpb - 1) / tpb);
    hipLaunchKernelGGL((dwpw_fused_kernel<MT, TW, NS, F16, NB>), dim3(grid), dim3(256 * NS), 0, s, x, packed, y, M, K, G, tiles_x,
                       tiles_y, ngrp, (unsigned)nwork, tpb, dbg);
    return gfn::check_launch("dwpw_fused_kernel");
}

// output channels: one workgroup computes all of them where they fit 2 slabs of <= 7 MFMA row tiles
// (M <= 448: every refiner), so the depthwise arithmetic of a cell tile is done once.  Narrow blocks
// (M <= 96: the fine scales, bound by LDS and HBM traffic rather than the matrix core) take 256-cell
// tiles with two output rows per depthwise thread when the map divides into them.
template <int TW, bool F16>
int launch_fused(const float *x, const float *packed, float *y, int B, int M, int K, int G, int dbg, hipStream_t s) {
    const int tiles = (M + 31) / 32;
    if (F16 && tiles <= 3 && TW >= 16 && G % (2 * kBN / TW) == 0) {  // fp32: the larger tiles cost a resident workgroup (LDS)
        if constexpr (F16 && TW >= 16) {
            switch (tiles) {
                case 1: return launch_fused_mt<1, TW, 1, F16, 2>(x, packed, y, B, M, K, G, dbg, s);
                case 2: return launch_fused_mt<2, TW, 1, F16, 2>(x, packed, y, B, M, K, G, dbg, s);
                default: return launch_fused_mt<3, TW, 1, F16, 2>(x, packed, y, B, M, K, G, dbg, s);
            }
        }
    }
    if (tiles <= 7) {
        switch (tiles) {
            case 1: return launch_fused_mt<1, TW, 1, F16, 1>(x, packed, y, B, M, K, G, dbg, s);
            case 2: return launch_fused_mt<2, TW, 1, F16, 1>(x, packed, y, B, M, K, G, dbg, s);
            case 3: return launch_fused_mt<3, TW, 1, F16, 1>(x, packed, y, B, M, K, G, dbg, s);
            case 4: return launch_fused_mt<4, TW, 1, F16, 1>(x, packed, y, B, M, K, G, dbg, s);
            case 5: return launch_fused_mt<5, TW, 1, F16, 1>(x, packed, y, B, M, K, G, dbg, s);
            case 6: return launch_fused_mt<6, TW, 1, F16, 1>(x, packed, y, B, M, K, G, dbg, s);
            default: return launch_fused_mt<7, TW, 1, F16, 1>(x, packed, y, B, M, K, G, dbg, s);
        }
    }
    const int ngrp = (tiles + 13) / 14;
    const int mt = ((tiles + ngrp - 1) / ngrp + 1) / 2;  // row tiles per slab
    switch (mt) {
        case 4: return launch_fused_mt<4, TW, 2, F16, 1>(x, packed, y, B, M, K, G, dbg, s);
        case 5: return launch_fused_mt<5, TW, 2, F16, 1>(x, packed, y, B, M, K, G, dbg, s);
        case 6: return launch_fused_mt<6, TW, 2, F16, 1>(x, packed, y, B, M, K, G, dbg, s);
        default: return launch_fused_mt<7, TW, 2, F16, 1>(x, packed, y, B, M, K, G, dbg, s);
    }
}

}  // namespace

GFN_EXPORT int64_t gfn_conv_block_packed_floats(int C, int M) {
    if (C <= 0 || M <= 0) return 0;
    return (int64_t)PackDims(C, M).total();
}

GFN_EXPORT int gfn_conv_block_pack(const float *dw_w, const float *dw_b, const float *bn_alpha, const float *bn_beta,
                                   const float *pw_w, const float *pw_b, float *packed, int C, int M, gfn_stream_t stream) {
    if (!dw_w || !bn_alpha || !bn_beta || !pw_w || !pw_b || !packed || C <= 0 || M <= 0)
        return gfn::fail(GFN_ERR_INVALID_ARG, "conv_block_pack: bad argument");
    hipLaunchKernelGGL(pack_block_kernel, dim3(grid_for((long)PackDims(C, M).total(), 1024)), dim3(256), 0, (hipStream_t)stream, dw_w,
                       dw_b, bn_alpha, bn_beta, pw_w, pw_b, packed, C, M);
    return gfn::check_launch("pack_block_kernel");
}

GFN_EXPORT int gfn_conv_block_fwd(const float *x, const float *packed, float *y, float *t_scratch, int B, int C, int M, int G,
                                  int variant, gfn_stream_t stream) {
    if (!x || !packed || !y || B < 0 || C <= 0 || M <= 0 || G <= 0) return gfn::fail(GFN_ERR_INVALID_ARG, "conv_block: bad argument");
    if (x == y) return gfn::fail(GFN_ERR_INVALID_ARG, "conv_block: in-place is not supported (cells read their neighbours)");
    if (B == 0) return GFN_OK;
    hipStream_t s = (hipStream_t)stream;
    if ((long)C * G * G > 0x7fffffffL) return gfn::fail(GFN_ERR_INVALID_ARG, "conv_block: C*G*G must fit 31 bits");
    const int dbg = variant >> 8;  // ablation mask, honoured by -DGFN_ABLATE builds only
#ifdef GFN_ABLATE
    variant &= 0xff;
#endif
    if (variant < 0 || variant > 3) return gfn::fail(GFN_ERR_INVALID_ARG, "conv_block: variant must be 0..3 (got %d)", variant);
    const bool f16 = (variant & 2) != 0;
    const bool fused = !(variant & 1) && (G & 3) == 0;
    if (fused) {
        // tile width: full 128-byte rows where the map allows, narrower tiles for the 5*2^k grids
        if (G % 32 == 0 || G > 160)
            return f16 ? launch_fused<32, true>(x, packed, y, B, M, C, G, dbg, s) : launch_fused<32, false>(x, packed, y, B, M, C, G, dbg, s);
        if (G % 16 == 0 || G > 64)
            return f16 ? launch_fused<16, true>(x, packed, y, B, M, C, G, dbg, s) : launch_fused<16, false>(x, packed, y, B, M, C, G, dbg, s);
        return f16 ? launch_fused<8, true>(x, packed, y, B, M, C, G, dbg, s) : launch_fused<8, false>(x, packed, y, B, M, C, G, dbg, s);
    }
    int nblk, mt;
    slab_shape(M, &nblk, &mt);
    if (!t_scratch) return gfn::fail(GFN_ERR_INVALID_ARG, "conv_block: the two-pass variant needs t_scratch (B*C*G*G floats)");
    if ((long)B * C > 0x7fffff || B > 65535) return gfn::fail(GFN_ERR_INVALID_ARG, "conv_block: batch too large for the two-pass variant");
    const int N = G * G;
    if ((G & 3) == 0) {
        const int bpp = (G * (G / 4) + 255) / 256;
        hipLaunchKernelGGL((dw5x5_kernel<4>), dim3((unsigned)(B * C * bpp)), dim3(256), 0, s, x, t_scratch, packed, C, G, bpp);
    } else {
        const int bpp = (N + 255) / 256;
        hipLaunchKernelGGL((dw5x5_kernel<1>), dim3((unsigned)(B * C * bpp)), dim3(256), 0, s, x, t_scratch, packed, C, G, bpp);
    }
    if (int rc = gfn::check_launch("dw5x5_kernel")) return rc;
    const dim3 grid((N + kBN - 1) / kBN, nblk, B), block(256);
#define GFN_PW(MT)                                                                                                        \
    if (f16)                                                                                                              \
        hipLaunchKernelGGL((pw_gemm_kernel<MT, true>), grid, block, 0, s, packed, (const float *)t_scratch, y, M, C, N);  \
    else                                                                                                                  \
        hipLaunchKernelGGL((pw_gemm_kernel<MT, false>), grid, block, 0, s, packed, (const float *)t_scratch, y, M, C, N)
    switch (mt) {
        case 1: GFN_PW(1); break;
        case 2: GFN_PW(2); break;
        case 3: GFN_PW(3); break;
        case 4: GFN_PW(4); break;
        case 5: GFN_PW(5); break;
        case 6: GFN_PW(6); break;
        default: GFN_PW(7); break;
    }
#undef GFN_PW
    return gfn::check_launch("pw_gemm_kernel");
}

GFN_EXPORT int gfn_pointwise_conv_fwd(const float *w, const float *bias, const float *t, float *y, int B, int M, int K, int N,
                                      gfn_stream_t stream) {
    if (!w || !bias || !t || !y || B < 0 || M <= 0 || K <= 0 || N <= 0) return gfn::fail(GFN_ERR_INVALID_ARG, "pointwise_conv: bad argument");
    if (M > 16) return gfn::fail(GFN_ERR_INVALID_ARG, "pointwise_conv: meant for a few output channels (M <= 16, got %d)", M);
    if (B == 0) return GFN_OK;
    hipLaunchKernelGGL(pw_small_kernel, dim3(grid_for((long)B * N)), dim3(256), 0, (hipStream_t)stream, w, t, bias, y, B, M, K, N);
    return gfn::check_launch("pw_small_kernel");
}
