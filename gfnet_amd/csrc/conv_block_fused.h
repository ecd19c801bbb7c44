// conv_block_fused.h -- the fused conv block (depthwise 5x5 + BatchNorm + ReLU + 1x1 conv in one kernel) of csrc/conv_stack.hip
// and csrc/conv_stack_half.hip (the same kernel on fp16 maps).  Opens its own anonymous namespace.
#pragma once
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
constexpr int kMM2 = 96;   // dwords per channel PAIR in the matrix-core depthwise parameters (see PackDims::mm_off)

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
    // matrix-core depthwise (fp16 maps): per channel pair kMM2 dwords -- [ch][dy][8] the taps of channel 2p+ch as fp16 in the
    // low (ch = 0) or high (ch = 1) half of a dword, entries 5..7 zero (what a lane whose window misses the tap reads), then
    // conv bias, alpha, beta of both channels as floats (80 .. 85)
    __host__ __device__ size_t mm_off() const { return wt16_off() + (size_t)Kp * Mp / 2; }
    __host__ __device__ size_t total() const { return mm_off() + (size_t)(Kp / 2) * kMM2; }
    // fp16 weights, in halfs from wt16_off: per K tile [k half kg][m][8], k = 16*kt + 8*kg + j: one 16-byte read = a lane's
    // A operand of v_mfma_f32_32x32x16_f16
    __host__ __device__ size_t wt16_index(int k, int m) const { return ((size_t)((k >> 4) * 2 + ((k >> 3) & 1)) * Mp + m) * 8 + (k & 7); }
    __host__ __device__ size_t wt_index(int k, int m) const {
        const int kt = k >> 4, r = k & 15, sg = r >> 3, kh = r & 1, j = (r & 7) >> 1;
        return ((size_t)((kt * 2 + sg) * 2 + kh) * Mp + m) * 4 + j;
    }
};

// The depthwise arithmetic, shared by both variants so that they agree bit for bit: 25 fmas per
// output in (dy, dx) order, then (acc + bias) * alpha + beta, relu.
__device__ __forceinline__ float dw_finish(float acc, float cb, float al, float be) { return fmaxf((acc + cb) * al + be, 0.f); }

// buffer addressing for the staging loads: a scalar descriptor per map / parameter block and one 32-bit offset per lane (the
// flat form cost ~15 VALU instructions of 64-bit address arithmetic per load: a quarter of a K tile's time in the wide blocks);
// reads past the end of the descriptor return 0 (channel pairs past C in the last K tile)
typedef __amdgpu_buffer_rsrc_t cb_rsrc_t;
__device__ __forceinline__ cb_rsrc_t cb_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 cb_ld4(cb_rsrc_t r, unsigned voff) {
    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float cb_ld(cb_rsrc_t r, unsigned voff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 0, 0));
}

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
// Maps in HBM: fp32 (B, C, G, G), or -- HIN / HOUT, fp16 maps -- (B, ceil(C/2), G, G) of half2 = channels (2p, 2p+1) of a
// cell side by side (the odd channel past C is zero): the order the halo is staged in and the order an accumulator lane
// holds its four consecutive output channels in, so neither side shuffles.
//
// MM (fp16 maps only): the depthwise 5x5 runs on the matrix core too.  A row of 8 output cells x 2 channels is a 16-cell x
// 2-channel window of the halo times a banded (Toeplitz) matrix of the pair's taps: v_mfma_f32_16x16x32_f16 with A = 16
// independent row segments of the halo (fp16 in LDS exactly as it sits in HBM: a lane's 8 k values are one 16-byte piece),
// B = the band of one tap row (built per pair from a 6-entry table), accumulated over the 5 tap rows.  85 % of the products
// are zeros, and it still costs a fifth of the VALU form (25 v_pk_fma_f32 per cell pair); BatchNorm + ReLU stay fp32 on the
// accumulator.  Taps rounded to fp16 (torch.autocast does the same to the conv weight).
//
// KW = 2 (wide blocks on fp16 maps): two 16-channel K tiles per iteration -- the wide blocks run one or two workgroups per CU
// through six short barrier-separated phases per K tile (3.2 k cycles, a third of it fixed latencies); 32 channels per
// iteration halve the barriers and the waits.  A block whose channel count leaves a single K tile for the last iteration
// reads zeros for the missing one (descriptor range checks on the map, the weights and the parameters).
template <int MT, int TW, int NS, bool F16, int NB, bool HIN = false, bool HOUT = false, bool MM = false, int KW = 1>
// (a fourth workgroup per CU for the narrowest blocks -- 128 VGPRs, 32 KB of LDS -- measured slower: 146/265 vs 143/218 us at
// C = 24, 256^2 / 320^2 maps; the memory system, not the CU, is what these blocks wait for)
__global__ __launch_bounds__(256 * NS, NS == 1 ? 2 : 1) void dwpw_fused_kernel(const float *__restrict__ x,
                                                                               const float *__restrict__ packed,
                                                                               float *__restrict__ y, int M, int K, int G,
                                                                               int tiles_x, int tiles_y, int ngrp, unsigned nwork,
                                                                               int tpb, int dbg_arg) {
    static_assert(F16 || !(HIN || HOUT), "fp16 maps go with fp16 1x1 operands");
    static_assert(!MM || F16, "the matrix-core depthwise takes fp16 operands");
    static_assert(KW == 1 || (KW == 2 && MM && F16), "two K tiles per iteration: matrix-core depthwise only");
    constexpr int NPT = kNP * KW, KTT = kKT * KW;  // channel pairs / channels per iteration
#ifdef GFN_ABLATE  // timing experiments only (tools/ablate_convblock.py): skip parts of the kernel; results are wrong
    const int dbg = dbg_arg;
    // bit 64: s_memtime stamps at the phase boundaries of a few workgroups (device printf at the end)
    const bool stamping = (dbg & 64) && (blockIdx.x % 997) == 500 && (threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) & 1) == 0;
    long long stamp[40];
    for (int i = 0; i < 40; ++i) stamp[i] = 0;
#define CSTAMP(i) do { if (stamping && (i) < 40) stamp[i] = __builtin_readcyclecounter(); } while (0)
#else
    constexpr int dbg = 0;
#define CSTAMP(i) do { } while (0)
#endif
    static_assert(NB == 1 || (NB == 2 && NS == 1 && MT <= 3), "256-cell tiles: one slab of at most 96 output channels");
    constexpr int NT = 256 * NS;
    constexpr int BN = kBN * NB;            // cells per workgroup tile
    constexpr int TH = BN / TW;             // tile rows
    constexpr int HR = TH + 4;              // halo rows
    constexpr int RV4 = (TW + 8) / 4;       // float4 per staged halo row and channel: cells col0-4 .. col0+TW+3
    constexpr int RPP = (TW + 8) * 2 + 4;   // LDS floats per halo row of a channel pair (+4: bank spread)
    constexpr int PP = HR * RPP;            // LDS floats per channel pair
    constexpr int PS = NPT * HR * RV4;      // staging slots: one = the same 4 cells of both channels of a pair
    constexpr int XPP = (PS + NT - 1) / NT;
    constexpr int BM = 32 * MT, BMS = BM * NS;
    constexpr int GP = F16 ? 2 : 4;         // 16-byte operand groups per weight row and K tile
    constexpr int AV4 = KW * GP * BMS;      // 16-byte pieces of one iteration's weight tile(s)
    constexpr int APT = (AV4 + NT - 1) / NT;
    constexpr int PW = MM ? kMM2 : kCP2;    // parameter dwords per channel pair
    constexpr int PPT = (NPT * PW + NT - 1) / NT;  // parameter dwords per thread and iteration
    // MM: the halo in LDS is fp16, one dword = the two channels of a cell; PH dwords per halo row -- the pitch that puts the 16
    // lanes of a ds_read_b128 group (row segment r = m % RS, chunk c = m / RS, piece kg) on distinct banks
    constexpr int PH = TW == 8 ? 24 : 48;
    constexpr int CHK = TW / 8, RS = 16 / CHK;  // 8-cell chunks per tile row, rows per 16-segment sub-tile
    static_assert(!MM || TH == RS * NB, "sub-tiles of 16 row segments");
    constexpr int CPT = 4 / NS;             // depthwise: cells per thread and row
    constexpr int RB = NB;                  //            rows per thread
    constexpr int TPP = kBN / CPT;          //            threads per channel pair
    constexpr int GPR = TW / CPT;           //            threads per tile row
    // Two-row depthwise threads step through the halo two rows at a time, and two row pitches are 0 mod 8 banks:
    // every other row PAIR is stored 16 bytes later (the pitch has the room), which puts the 16 lanes of one
    // ds_read_b128 group back on all 32 banks (measured: 69% of LDS cycles were bank conflicts without it).
    constexpr bool SWZ = NB == 2;

    __shared__ __attribute__((aligned(16))) float Xs[MM ? NPT * HR * PH : kNP * PP];
    // fp32: [buf][(sg*2+kh)*BMS + m] = A operands of k-steps 4sg .. 4sg+3;  fp16: [buf][kg*BMS + m] = 8 halfs k = 8kg ..
    __shared__ float4 As4[2][KW * GP * BMS];
    // B operand tile: fp32 [channel k][cell]; fp16 [pair][cell] of half2 (channels 2p, 2p+1) -- consecutive depthwise
    // threads write consecutive 16-byte pieces, the matrix lanes read consecutive dwords
    __shared__ __attribute__((aligned(16))) float Bs[F16 ? NPT * BN : kKT * BN];
    __shared__ __attribute__((aligned(16))) float Ps[NPT * PW];
    // The 1x1 bias of the item's output slabs, by item parity.  It rides the load pipeline (fetched with an item's first K tile,
    // filed at that tile's commit): fetched in the epilogue its wait would be vmcnt(0) -- on gfx9 that also waits for the next
    // item's prefetch and for the previous accumulator tile's stores to be acknowledged (measured: the store phase cost as
    // much as the depthwise).
    __shared__ __attribute__((aligned(16))) float Bias_s[2][BMS];

    const PackDims pd(K, M);
    const float *cp = packed + (MM ? pd.mm_off() : pd.cp_off());
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
    const int nk = (Kp / kKT + KW - 1) / KW;
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
        xl[i] = MM ? p * (HR * PH) + hr * PH + 4 * q : p * PP + hr * RPP + 8 * q + (SWZ && ((hr >> 1) & 1) ? 4 : 0);
    }
    unsigned l_item = w_begin;  // load stage: the (item, K tile) the next issue() fetches
    int l_kt = 0, l_m0 = 0, l_valid = 0;  // l_valid bit i: slot i lies inside the map (else zero padding)
    const unsigned map_bytes = (unsigned)(HIN ? (K + 1) / 2 : K) * (unsigned)plane * 4u;  // HIN: dwords = half2 cells
    cb_rsrc_t x_rsrc = cb_rsrc(x, map_bytes);
    unsigned av[APT > 0 ? APT : 1];  // byte offsets of this thread's weight-tile pieces inside a K tile's block
    auto load_stage_enter_item = [&]() {
        int b, row0, col0;
        decode(l_item, b, row0, col0, l_m0);
        x_rsrc = cb_rsrc(x + (size_t)b * (HIN ? (K + 1) / 2 : K) * plane, map_bytes);
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int e = tid + NT * i;
            const int g = e / BMS, m = l_m0 + e - g * BMS;
            const bool ok = e < AV4 && m < Mp;
            av[i] = (unsigned)((ok ? g : 0) * Mp + (ok ? m : 0)) * 16u;
        }
        l_valid = 0;
#pragma unroll
        for (int i = 0; i < XPP; ++i) {
            const int e = tid + NT * i;
            const int p = e / (HR * RV4), rem = e - p * (HR * RV4);
            const int hr = rem / RV4, q = rem - hr * RV4;
            const int gy = row0 - 2 + hr, gx = col0 - 4 + 4 * q;
            const bool ok = e < PS && (unsigned)gy < (unsigned)G && gx >= 0 && gx < G;  // G % 4 == 0: 4 cells in or out together
            // byte offset inside the map of this slot's piece in the K tile's first channel pair (HIN) / channel
            xg[i] = ((ok ? gy * G + gx : 0) + (HIN ? xp2[i] >> 1 : xp2[i]) * plane) * 4;
            l_valid |= ok ? 1 << i : 0;
        }
    };
    if (!(dbg & 32))
        for (int e = tid; e < (MM ? NPT * HR * PH : kNP * PP) / 4; e += NT) reinterpret_cast<float4 *>(Xs)[e] = make_float4(0.f, 0.f, 0.f, 0.f);

    static_assert(APT <= 4, "weight tile slots");
    static_assert(BMS <= NT, "one bias value per thread");
    float4 xr0[XPP], xr1[HIN ? 1 : XPP];
    float4 ar0, ar1, ar2, ar3;  // named, not an array: the compiler demotes a float4 array here to LDS
    float pr[PPT];
    int r_valid = 0;  // l_valid of the item in the registers; bit 8: its first K tile (refresh the zero padding); bit 9: a first K
                      // tile (file the bias), bit 10: the item's parity
    float bias_r = 0.f;
    const cb_rsrc_t a_rsrc = cb_rsrc(F16 ? packed + pd.wt16_off() : wt, (unsigned)Kp * (unsigned)Mp * (F16 ? 2u : 4u));
    const cb_rsrc_t p_rsrc = cb_rsrc(cp, (unsigned)(Kp / 2) * PW * 4u);
    auto a_load = [&](int kt, int i) { return cb_ld4(a_rsrc, av[i] + (unsigned)(kt * KW * GP * Mp) * 16u); };
    auto a_store = [&](int buf, int i, const float4 &v) {
        const int e = tid + NT * i;
        if (e < AV4) As4[buf][e] = v;
    };
    auto issue = [&]() {  // fetch (l_item, l_kt) into registers, advance the load stage
        if (dbg & 1) return;
        const unsigned kx = (unsigned)l_kt * (unsigned)((HIN ? NPT : KTT) * plane) * 4u;  // the iteration's first pair / channel
#pragma unroll
        for (int i = 0; i < XPP; ++i) {
            xr0[i] = cb_ld4(x_rsrc, (unsigned)xg[i] + kx);  // HIN: 4 cells of a channel pair; past C: zeros (their taps are 0 too)
            if constexpr (!HIN) xr1[i] = cb_ld4(x_rsrc, (unsigned)xg[i] + kx + (unsigned)plane * 4u);
        }
        if constexpr (APT > 0) ar0 = a_load(l_kt, 0);
        if constexpr (APT > 1) ar1 = a_load(l_kt, 1);
        if constexpr (APT > 2) ar2 = a_load(l_kt, 2);
        if constexpr (APT > 3) ar3 = a_load(l_kt, 3);
#pragma unroll
        for (int i = 0; i < PPT; ++i) pr[i] = cb_ld(p_rsrc, (unsigned)(l_kt * NPT * PW + min(tid + NT * i, NPT * PW - 1)) * 4u);
        r_valid = l_valid | (l_kt == 0 && l_item != w_begin ? 256 : 0);
        if (l_kt == 0) {
            r_valid |= 512 | (((l_item - w_begin) & 1u) ? 1024 : 0);
            const int m = l_m0 + tid;
            bias_r = bias[tid < BMS && m < Mp ? m : 0];  // padded to Mp
        }
        if (++l_kt == nk) {
            l_kt = 0;
            if (++l_item < w_end) load_stage_enter_item();
        }
    };
    auto commit = [&](int buf) {
        if (dbg & 16) return;
#pragma unroll
        for (int i = 0; i < XPP; ++i) {
            if constexpr (MM) {
                if (r_valid & (1 << i)) {
                    f32x4 v;
                    if constexpr (HIN) {
                        v = f32x4{xr0[i].x, xr0[i].y, xr0[i].z, xr0[i].w};
                    } else {  // the fp32 concat: round to fp16 here (what autocast does to the first conv's input)
                        const f16x8 h = {(_Float16)xr0[i].x, (_Float16)xr1[i].x, (_Float16)xr0[i].y, (_Float16)xr1[i].y,
                                         (_Float16)xr0[i].z, (_Float16)xr1[i].z, (_Float16)xr0[i].w, (_Float16)xr1[i].w};
                        v = __builtin_bit_cast(f32x4, h);
                    }
                    *reinterpret_cast<f32x4 *>(&Xs[xl[i]]) = v;
                } else if ((r_valid & 256) && tid + NT * i < PS) {
                    *reinterpret_cast<float4 *>(&Xs[xl[i]]) = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                continue;
            }
            if (r_valid & (1 << i)) {
                f32x4 lo, hi;
                if constexpr (HIN) {
                    const f16x8 h = __builtin_bit_cast(f16x8, xr0[i]);
                    lo = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
                    hi = f32x4{(float)h[4], (float)h[5], (float)h[6], (float)h[7]};
                } else {
                    lo = f32x4{xr0[i].x, xr1[i].x, xr0[i].y, xr1[i].y};
                    hi = f32x4{xr0[i].z, xr1[i].z, xr0[i].w, xr1[i].w};
                }
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
        for (int i = 0; i < PPT; ++i)
            if (PPT * NT == NPT * PW || tid + NT * i < NPT * PW) Ps[tid + NT * i] = pr[i];
        if ((r_valid & 512) && tid < BMS) Bias_s[(r_valid >> 10) & 1][tid] = bias_r;
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
    CSTAMP(0);
    load_stage_enter_item();
    issue();
    CSTAMP(1);
    __syncthreads();  // Xs zeroed
    CSTAMP(2);
    unsigned c_item = w_begin;  // the item being accumulated
    int c_kt = 0;
    for (int t = 0; t < total; ++t) {
        const int buf = t & 1;
        commit(buf);
        CSTAMP(3 + 6 * t);
        __syncthreads();
        CSTAMP(4 + 6 * t);
        if (t + 1 < total) issue();
        CSTAMP(5 + 6 * t);
        if constexpr (MM) {
            if (!(dbg & 2)) {
                typedef float f32x4v __attribute__((ext_vector_type(4)));
                const int n = lane & 15, xo = n >> 1, ch = n & 1, kg = lane >> 4, mrow = lane & 15;
                const int ar = mrow % RS, ac = mrow / RS;  // A operand: this lane's row segment
                const float *xbase = &Xs[ar * PH + 8 * ac + 4 * kg];
                _Float16 *bs16 = reinterpret_cast<_Float16 *>(Bs);
                int tix[4];  // table entry of this lane's k pair i (cell 4kg + i of the window) for its output cell xo; 5 = zero
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int dx = 4 * kg + i - xo - 2;
                    tix[i] = ch * 40 + ((unsigned)dx <= 4u ? dx : 5);
                }
#pragma unroll
                for (int pi = 0; pi < NPT / (4 * NS); ++pi) {
                    const int pr_ = wave + 4 * NS * pi;  // channel pair of the K tile
                    const float *tw = &Ps[pr_ * kMM2];
                    f16x8 T[5];
#pragma unroll
                    for (int dy = 0; dy < 5; ++dy) {
                        const f32x4v t = {tw[tix[0] + 8 * dy], tw[tix[1] + 8 * dy], tw[tix[2] + 8 * dy], tw[tix[3] + 8 * dy]};
                        T[dy] = __builtin_bit_cast(f16x8, t);
                    }
                    const float cb = tw[80 + ch], al = tw[82 + ch], be = tw[84 + ch];
#pragma unroll
                    for (int sub = 0; sub < NB; ++sub) {
                        f32x4v d = {0.f, 0.f, 0.f, 0.f};
                        const float *xs = xbase + pr_ * (HR * PH) + sub * RS * PH;
#pragma unroll
                        for (int dy = 0; dy < 5; ++dy) {
                            const f16x8 a = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4v *>(xs + dy * PH));
                            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, T[dy], d, 0, 0, 0);
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {  // D row 4kg + j = row segment (r, c); column n = (cell xo, channel ch)
                            const int m = 4 * kg + j, r = m % RS, c = m / RS;
                            const int cell = (sub * RS + r) * TW + 8 * c + xo;
                            bs16[(pr_ * BN + cell) * 2 + ch] = (_Float16)dw_finish(d[j], cb, al, be);
                        }
                    }
                }
            }
        } else
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
        CSTAMP(6 + 6 * t);
        __syncthreads();
        CSTAMP(7 + 6 * t);
#pragma unroll
        for (int g = 0; g < NB; ++g) {
            if (dbg & 4) break;
            const int cell = (g * 4 + cw) * 32 + col;  // this lane's B column
            if constexpr (F16) {  // one v_mfma_f32_32x32x16_f16 per row tile: lane (n or m = lane&31, kg = lane>>5) holds 8 halfs
                // channel pairs 4kh .. 4kh+3 of this cell = halfs k = 8kh .. 8kh+7 (of k-step ks)
#pragma unroll
                for (int ks = 0; ks < KW; ++ks) {
                    const float *bp = &Bs[(8 * ks + 4 * kh) * BN + cell];
                    const f32x4 bq = {bp[0], bp[BN], bp[2 * BN], bp[3 * BN]};
                    const f16x8 bv = __builtin_bit_cast(f16x8, bq);
                    const f16x8 *asrc = reinterpret_cast<const f16x8 *>(&As4[buf][(ks * GP + kh) * BMS + slab * BM + col]);
                    f16x8 avv[MT];
#pragma unroll
                    for (int i = 0; i < MT; ++i) avv[i] = asrc[i * 32];
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[g * MT + i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(avv[i], bv, acc[g * MT + i], 0, 0, 0);
                }
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
        CSTAMP(8 + 6 * t);
        if (++c_kt < nk) continue;
        // item finished: D[row][col], col = lane&31 -> cell (g*4+cw)*32+col of the tile, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
        int b, row0, col0, m0;
        decode(c_item, b, row0, col0, m0);
        const int bpar = (int)((c_item - w_begin) & 1u);
        c_kt = 0;
        ++c_item;
#pragma unroll
    for (int g = 0; g < NB; ++g) {
        const int p = (g * 4 + cw) * 32 + col;
        const int gy = row0 + p / TW, gx = col0 + p % TW;
        if (gy >= G || gx >= G || ((dbg & 8) && acc[0][0] != 12345.f)) continue;
        float *yb = y + (size_t)b * (HOUT ? (M + 1) / 2 : M) * plane + (size_t)gy * G + gx;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int mb = m0 + slab * BM + i * 32;  // 32 output channels of this accumulator tile
            if (mb >= M) break;
            float4 bq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const float4 *>(&Bias_s[bpar][slab * BM + i * 32 + 8 * q + 4 * kh]);
            if constexpr (HOUT) {  // channels m .. m+3 of this lane's cell = two half2 dwords, 128 bytes per 32 lanes; rows past M are 0
                float *yt = yb + (size_t)((mb + 4 * kh) >> 1) * plane;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = mb + 8 * q + 4 * kh;
                    const f16x2 h0 = {(_Float16)(acc[g * MT + i][4 * q] + bq[q].x), (_Float16)(acc[g * MT + i][4 * q + 1] + bq[q].y)};
                    const f16x2 h1 = {(_Float16)(acc[g * MT + i][4 * q + 2] + bq[q].z), (_Float16)(acc[g * MT + i][4 * q + 3] + bq[q].w)};
                    // streaming stores where a wave's 32 lanes cover whole 64/128-byte lines: measured 5-12 % at C = 73 .. 417, a
                    // loss on the 8-cell-wide tiles (32-byte row segments) and at C = 24 (one slab: 149/246 vs 143/218 us)
                    if constexpr (TW >= 16 && MT >= 2) {
                        if (m < M) __builtin_nontemporal_store(__builtin_bit_cast(float, h0), &yt[(size_t)(4 * q) * plane]);
                        if (m + 2 < M) __builtin_nontemporal_store(__builtin_bit_cast(float, h1), &yt[(size_t)(4 * q + 1) * plane]);
                    } else {
                        if (m < M) yt[(size_t)(4 * q) * plane] = __builtin_bit_cast(float, h0);
                        if (m + 2 < M) yt[(size_t)(4 * q + 1) * plane] = __builtin_bit_cast(float, h1);
                    }
                }
                continue;
            }
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
        CSTAMP(33 + (int)(c_item - w_begin));  // item stored
    }
#ifdef GFN_ABLATE
    if (stamping) {
        for (int i = 1; i < 40; ++i)
            if (stamp[i]) printf("CS %u %d %d %lld\n", blockIdx.x, (int)(threadIdx.x >> 6), i, stamp[i] - stamp[0]);
    }
#endif
#undef CSTAMP
}

// GFN_CONV_TPB (environment, experiments): work items per workgroup of the fused kernel, 0 = heuristic
static int g_conv_tpb = [] {
    const char *e = gfn::exp_env("GFN_CONV_TPB");
    return e ? atoi(e) : 0;
}();

template <int MT, int TW, int NS, bool F16, int NB, bool HIN = false, bool HOUT = false, bool MM = false, int KW = 1>
int launch_fused_mt(const float *x, const float *packed, float *y, int B, int M, int K, int G, int dbg, hipStream_t s) {
    constexpr int TH = kBN * NB / TW;
    const int tiles_x = (G + TW - 1) / TW, tiles_y = (G + TH - 1) / TH;
    const int ngrp = (M + 32 * MT * NS - 1) / (32 * MT * NS);
    const long nwork = (long)B * tiles_x * tiles_y * ngrp;
    if (nwork > 0x7fffffffL) return gfn::fail(GFN_ERR_INVALID_ARG, "conv_block: too many tiles");
    // work items per workgroup (pipelined back to back)
    int tpb = nwork >= 16384 ? 2 : 1;  // measured: pays only on the largest grids
#ifdef GFN_CONV_TPB_FORCE  // A/B builds (python -m gfnet_amd.build --variant NAME --src conv_stack_half.hip -DGFN_CONV_TPB_FORCE=4)
    if (nwork >= 16384) tpb = GFN_CONV_TPB_FORCE;
#endif
    if (g_conv_tpb > 0) tpb = g_conv_tpb;
    const unsigned grid = (unsigned)((nwork + tpb - 1) / tpb);
    hipLaunchKernelGGL((dwpw_fused_kernel<MT, TW, NS, F16, NB, HIN, HOUT, MM, KW>), dim3(grid), dim3(256 * NS), 0, s, x, packed, y, M, K, G, tiles_x,
                       tiles_y, ngrp, (unsigned)nwork, tpb, dbg);
    return gfn::check_launch("dwpw_fused_kernel");
}

// output channels: one workgroup computes all of them where they fit 2 slabs of <= 7 MFMA row tiles
// (M <= 448: every refiner), so the depthwise arithmetic of a cell tile is done once.  Narrow blocks
// (M <= 96: the fine scales, bound by LDS and HBM traffic rather than the matrix core) take 256-cell
// tiles with two output rows per depthwise thread when the map divides into them.
template <int TW, bool F16, bool HIN = false, bool HOUT = false, bool MM = false, int KW = 1>
int launch_fused(const float *x, const float *packed, float *y, int B, int M, int K, int G, int dbg, hipStream_t s) {
    const int tiles = (M + 31) / 32;
    static const bool nb1 = gfn::exp_env("GFN_CONV_NB1") != nullptr;  // experiments: 128-cell tiles for the narrow blocks too
    if (F16 && tiles <= 3 && TW >= 16 && G % (2 * kBN / TW) == 0 && !nb1) {  // fp32: the larger tiles cost a resident workgroup (LDS)
        if constexpr (F16 && TW >= 16) {
            switch (tiles) {
                case 1: return launch_fused_mt<1, TW, 1, F16, 2, HIN, HOUT, MM>(x, packed, y, B, M, K, G, dbg, s);
                case 2: return launch_fused_mt<2, TW, 1, F16, 2, HIN, HOUT, MM>(x, packed, y, B, M, K, G, dbg, s);
                default: return launch_fused_mt<3, TW, 1, F16, 2, HIN, HOUT, MM>(x, packed, y, B, M, K, G, dbg, s);
            }
        }
    }
    static const bool ns1 = gfn::exp_env("GFN_CONV_NS1") != nullptr;  // experiments: one slab per workgroup, two workgroups per cell tile
    if (tiles <= 7 || ns1) {
        switch (tiles) {
            case 1: return launch_fused_mt<1, TW, 1, F16, 1, HIN, HOUT, MM>(x, packed, y, B, M, K, G, dbg, s);
            case 2: return launch_fused_mt<2, TW, 1, F16, 1, HIN, HOUT, MM>(x, packed, y, B, M, K, G, dbg, s);
            case 3: return launch_fused_mt<3, TW, 1, F16, 1, HIN, HOUT, MM>(x, packed, y, B, M, K, G, dbg, s);
            case 4: return launch_fused_mt<4, TW, 1, F16, 1, HIN, HOUT, MM, KW>(x, packed, y, B, M, K, G, dbg, s);
            case 5: return launch_fused_mt<5, TW, 1, F16, 1, HIN, HOUT, MM, KW>(x, packed, y, B, M, K, G, dbg, s);
            case 6: return launch_fused_mt<6, TW, 1, F16, 1, HIN, HOUT, MM, KW>(x, packed, y, B, M, K, G, dbg, s);
            default: return launch_fused_mt<7, TW, 1, F16, 1, HIN, HOUT, MM, KW>(x, packed, y, B, M, K, G, dbg, s);
        }
    }
    const int ngrp = (tiles + 13) / 14;
    const int mt = ((tiles + ngrp - 1) / ngrp + 1) / 2;  // row tiles per slab
    switch (mt) {
        case 4: return launch_fused_mt<4, TW, 2, F16, 1, HIN, HOUT, MM, KW>(x, packed, y, B, M, K, G, dbg, s);
        case 5: return launch_fused_mt<5, TW, 2, F16, 1, HIN, HOUT, MM, KW>(x, packed, y, B, M, K, G, dbg, s);
        case 6: return launch_fused_mt<6, TW, 2, F16, 1, HIN, HOUT, MM, KW>(x, packed, y, B, M, K, G, dbg, s);
        default: return launch_fused_mt<7, TW, 2, F16, 1, HIN, HOUT, MM, KW>(x, packed, y, B, M, K, G, dbg, s);
    }
}

}  // namespace
