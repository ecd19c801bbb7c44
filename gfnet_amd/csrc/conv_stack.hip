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

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kKT = 16;    // channels per K tile
constexpr int kBN = 128;   // cells per workgroup: 4 waves x 32
constexpr int kCP = 48;    // floats per channel in the packed depthwise parameters: 5 tap rows of 8 (5 used), bias, alpha, beta at 40..42

__host__ __device__ inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// ---- packed parameters of one block ---------------------------------------------------------------
// [ cp: Kp x 48 ][ wt: Kp x Mp (W transposed, zero padded) ][ bias: Mp ],  Kp = ceil16(C), Mp = ceil32(M)
struct PackDims {
    int Kp, Mp;
    __host__ __device__ PackDims(int C, int M) : Kp(round_up(C, kKT)), Mp(round_up(M, 32)) {}
    __host__ __device__ size_t cp_off() const { return 0; }
    __host__ __device__ size_t wt_off() const { return (size_t)Kp * kCP; }
    __host__ __device__ size_t bias_off() const { return wt_off() + (size_t)Kp * Mp; }
    __host__ __device__ size_t total() const { return bias_off() + Mp; }
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
            const int k = (int)(i / kCP), j = (int)(i % kCP);
            if (k < C) {
                if (j < 40) v = (j & 7) < 5 ? dw_w[(size_t)k * 25 + (j >> 3) * 5 + (j & 7)] : 0.f;
                else if (j == 40) v = dw_b ? dw_b[k] : 0.f;
                else if (j == 41) v = alpha[k];
                else if (j == 42) v = beta[k];
            }
        } else if (i < d.bias_off()) {
            const size_t e = i - d.wt_off();
            const int k = (int)(e / d.Mp), m = (int)(e % d.Mp);
            if (k < C && m < M) v = pw_w[(size_t)m * C + k];
        } else {
            const int m = (int)(i - d.bias_off());
            if (m < M) v = pw_b[m];
        }
        packed[i] = v;
    }
}

// The depthwise arithmetic, shared by both variants so that they agree bit for bit: 25 fmas per
// output in (dy, dx) order, then (acc + bias) * alpha + beta, relu.
__device__ __forceinline__ float dw_finish(float acc, float cb, float al, float be) { return fmaxf((acc + cb) * al + be, 0.f); }

// ---- fused block ------------------------------------------------------------------------------------
// Persistent workgroups walk a contiguous run of work items (cell tile x output slab); the (item,
// K tile) pairs form one software pipeline:  global loads run three steps ahead (registers), the LDS
// commit two, the depthwise arithmetic one step ahead of the matrix-core step it feeds, so every wave
// interleaves VALU (depthwise for step i+1) with MFMA (step i) and no load latency is exposed.
template <int TW>
struct FusedGeom {
    static constexpr int TH = kBN / TW;       // tile rows
    static constexpr int HR = TH + 4;         // halo rows
    static constexpr int RV4 = (TW + 8) / 4;  // float4 per staged halo row: columns col0-4 .. col0+TW+3
    static constexpr int RP = TW + 8 + 4;     // LDS row pitch (floats): +4 spreads the b128 reads of a 16-lane group
    static constexpr int CPITCH = HR * RP;
    static constexpr int XV4 = kKT * HR * RV4;  // float4 of one halo stage
    static constexpr int XPT = (XV4 + 255) / 256;
    static constexpr int OPR = TW / 8;        // 8-cell groups per tile row
};

template <int MT, int TW>
__global__ __launch_bounds__(256, 2) void dwpw_fused_kernel(const float *__restrict__ x, const float *__restrict__ packed,
                                                            float *__restrict__ y, int M, int K, int G, int tiles_x, int tiles_y,
                                                            int nblk, unsigned nwork) {
    using Geo = FusedGeom<TW>;
    constexpr int TH = Geo::TH, HR = Geo::HR, RV4 = Geo::RV4, RP = Geo::RP, CPITCH = Geo::CPITCH, XV4 = Geo::XV4, XPT = Geo::XPT,
                  OPR = Geo::OPR;
    constexpr int BM = 32 * MT;
    constexpr int AV4 = kKT * BM / 4;
    constexpr int APT = (AV4 + 255) / 256;
    constexpr int PPT = kKT * kCP / 256;  // parameter floats per thread and stage

    __shared__ __attribute__((aligned(16))) float Xs[kKT * CPITCH];
    __shared__ __attribute__((aligned(16))) float As[2][kKT][BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][kKT][kBN];
    __shared__ __attribute__((aligned(16))) float Ps[kKT * kCP];

    const PackDims pd(K, M);
    const float *cp = packed + pd.cp_off();
    const float *wt = packed + pd.wt_off();
    const float *bias = packed + pd.bias_off();
    const int Mp = pd.Mp, nk = pd.Kp / kKT;
    const size_t plane = (size_t)G * G;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned lb = gfn::xcd_remap(blockIdx.x, gridDim.x);
    const unsigned w_begin = (unsigned)(((unsigned long long)lb * nwork) / gridDim.x);
    const unsigned w_end = (unsigned)(((unsigned long long)(lb + 1) * nwork) / gridDim.x);
    const int total = (int)(w_end - w_begin) * nk;  // pipeline steps of this workgroup
    if (total <= 0) return;

    auto decode = [&](unsigned item, int &b, int &row0, int &col0, int &m0) {
        const unsigned mblk = item % (unsigned)nblk;
        item /= (unsigned)nblk;
        const unsigned tx = item % (unsigned)tiles_x;
        item /= (unsigned)tiles_x;
        const unsigned ty = item % (unsigned)tiles_y;
        b = (int)(item / (unsigned)tiles_y);
        row0 = (int)ty * TH, col0 = (int)tx * TW, m0 = (int)mblk * BM;
    };

    // ---- load stage state: which (item, K tile) the next issue() fetches
    unsigned l_item = w_begin;
    int l_kt = 0, l_m0 = 0;
    const float *l_xb = x;
    int xoff[XPT];
    bool xok[XPT];
    auto load_stage_enter_item = [&]() {
        int b, row0, col0;
        decode(l_item, b, row0, col0, l_m0);
        l_xb = x + (size_t)b * K * plane;
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int e = tid + 256 * i;
            const int ch = e / (HR * RV4), rem = e - ch * (HR * RV4);
            const int hr = rem / RV4, q = rem - hr * RV4;
            const int gy = row0 - 2 + hr, gx = col0 - 4 + 4 * q;
            const bool ok = e < XV4 && (unsigned)gy < (unsigned)G && gx >= 0 && gx < G;  // G % 4 == 0: whole float4 in or out
            xok[i] = ok;
            xoff[i] = ok ? gy * G + gx : 0;
        }
    };
    float4 xr[XPT], ar[APT];
    float pr[PPT];
    auto issue = [&]() {  // fetch (l_item, l_kt) into registers, then advance the load stage
        const int k0 = l_kt * kKT;
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int e = tid + 256 * i;
            const int kk = k0 + e / (HR * RV4);
            const bool ok = xok[i] && kk < K;
            const float4 v = *reinterpret_cast<const float4 *>(l_xb + (size_t)(ok ? kk : 0) * plane + xoff[i]);
            xr[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int e = tid + 256 * i;
            const int k = e / (BM / 4), m4 = e - k * (BM / 4);
            const int m = l_m0 + 4 * m4;
            const bool ok = e < AV4 && m < Mp;
            const float4 v = *reinterpret_cast<const float4 *>(wt + (size_t)(k0 + (ok ? k : 0)) * Mp + (ok ? m : 0));
            ar[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < PPT; ++i) pr[i] = cp[(size_t)k0 * kCP + tid + 256 * i];
        if (++l_kt == nk) {
            l_kt = 0;
            if (++l_item < w_end) load_stage_enter_item();
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int e = tid + 256 * i;
            const int ch = e / (HR * RV4), rem = e - ch * (HR * RV4);
            const int hr = rem / RV4, q = rem - hr * RV4;
            if (e < XV4) *reinterpret_cast<float4 *>(&Xs[ch * CPITCH + hr * RP + 4 * q]) = xr[i];
        }
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int e = tid + 256 * i;
            const int k = e / (BM / 4), m4 = e - k * (BM / 4);
            if (e < AV4) *reinterpret_cast<float4 *>(&As[buf][k][4 * m4]) = ar[i];
        }
#pragma unroll
        for (int i = 0; i < PPT; ++i) Ps[tid + 256 * i] = pr[i];
    };

    const int col = lane & 31, kh = lane >> 5;
    // depthwise role: channel dk of the K tile, tile row dr, 8-cell group dg
    const int dk = tid >> 4, dro = tid & 15, dr = dro / OPR, dg = dro - dr * OPR;
    const float *dw_src = &Xs[dk * CPITCH + dr * RP + 8 * dg];  // halo columns 8dg .. 8dg+15 = cells 8dg-4 .. 8dg+11
    const float *dw_par = &Ps[dk * kCP];
    const int dw_dst = dk * kBN + dr * TW + 8 * dg;

    // depthwise 5x5 + affine + relu for the staged K tile: 8 cells of one row of one channel into Bs[buf]
    float a8[8];
    auto dw_row = [&](int dy) {
        const float4 w0 = *reinterpret_cast<const float4 *>(dw_par + 8 * dy);
        const float w4 = dw_par[8 * dy + 4];
        const float w[5] = {w0.x, w0.y, w0.z, w0.w, w4};
        float v[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 f = *reinterpret_cast<const float4 *>(dw_src + dy * RP + 4 * q);
            v[4 * q] = f.x, v[4 * q + 1] = f.y, v[4 * q + 2] = f.z, v[4 * q + 3] = f.w;
        }
#pragma unroll
        for (int dx = 0; dx < 5; ++dx)
#pragma unroll
            for (int j = 0; j < 8; ++j) a8[j] = fmaf(w[dx], v[j + dx + 2], a8[j]);
    };
    auto dw_store = [&](int buf) {
        const float cb = dw_par[40], al = dw_par[41], be = dw_par[42];
        float4 o0, o1;
        o0.x = dw_finish(a8[0], cb, al, be), o0.y = dw_finish(a8[1], cb, al, be);
        o0.z = dw_finish(a8[2], cb, al, be), o0.w = dw_finish(a8[3], cb, al, be);
        o1.x = dw_finish(a8[4], cb, al, be), o1.y = dw_finish(a8[5], cb, al, be);
        o1.z = dw_finish(a8[6], cb, al, be), o1.w = dw_finish(a8[7], cb, al, be);
        float *dst = &Bs[buf][0][0] + dw_dst;
        *reinterpret_cast<float4 *>(dst) = o0;
        *reinterpret_cast<float4 *>(dst + 4) = o1;
    };

    f32x16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // ---- prologue: step 0 staged and its depthwise done, step 1 staged, step 2 in registers
    load_stage_enter_item();
    issue();
    commit(0);
    __syncthreads();
    if (total > 1) issue();
#pragma unroll
    for (int j = 0; j < 8; ++j) a8[j] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 5; ++dy) dw_row(dy);
    dw_store(0);
    __syncthreads();
    if (total > 1) commit(1);
    if (total > 2) issue();
    __syncthreads();

    unsigned c_item = w_begin;  // MFMA stage
    int c_kt = 0;
    for (int it = 0; it < total; ++it) {
        const int buf = it & 1;
        // MFMA for step it (As[buf], Bs[buf]) interleaved with the depthwise of step it+1 (Xs -> Bs[buf^1]);
        // past the last step the depthwise runs on stale data and its output is never read
#pragma unroll
        for (int j = 0; j < 8; ++j) a8[j] = 0.f;
#pragma unroll
        for (int s = 0; s < kKT / 2; ++s) {
            const float bv = Bs[buf][2 * s + kh][wave * 32 + col];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const float av = As[buf][2 * s + kh][i * 32 + col];
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
            }
            if (s < 5) dw_row(s);
        }
        dw_store(buf ^ 1);
        if (++c_kt == nk) {  // item finished: D[row][col], col = lane&31 -> cell wave*32+col, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
            int b, row0, col0, m0;
            decode(c_item, b, row0, col0, m0);
            const int p = wave * 32 + col;
            const int gy = row0 + p / TW, gx = col0 + p % TW;
            if (gy < G && gx < G) {
                float *yb = y + (size_t)b * M * plane + (size_t)gy * G + gx;
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                        if (m < M) yb[(size_t)m * plane] = acc[i][r] + bias[m];
                    }
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            c_kt = 0;
            ++c_item;
        }
        __syncthreads();
        if (it + 2 < total) commit(buf);
        if (it + 3 < total) issue();
        __syncthreads();
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
    const float *wc = packed + (size_t)c * kCP;
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
                for (int q = 0; q < 4; ++q) acc[q] = fmaf(wc[dy * 8 + dx], v[q + dx + 2], acc[q]);
        } else {
#pragma unroll
            for (int dx = 0; dx < 5; ++dx) {
                const int xx = jv + dx - 2;
                const bool ok = rok && (unsigned)xx < (unsigned)G;
                const float xv = row[ok ? xx : 0];
                acc[0] = fmaf(wc[dy * 8 + dx], ok ? xv : 0.f, acc[0]);
            }
        }
    }
    float *dst = t + (size_t)pl * G * G + (size_t)i * G + (size_t)jv * VEC;
#pragma unroll
    for (int q = 0; q < VEC; ++q) dst[q] = dw_finish(acc[q], wc[40], wc[41], wc[42]);
}

// y[b] = W . t[b] + bias on the fp32 matrix core; same k order as the fused kernel.
template <int MT>
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
        for (int e = tid; e < kKT * (BM / 4); e += 256) {
            const int k = e / (BM / 4), m4 = e - k * (BM / 4);
            const int m = m0 + m4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < Mp) v = *reinterpret_cast<const float4 *>(wt + (size_t)(k0 + k) * Mp + m);
            *reinterpret_cast<float4 *>(&As[k][m4 * 4]) = v;
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
#pragma unroll
        for (int s = 0; s < kKT / 2; ++s) {
            const float bv = Bs[2 * s + kh][wave * 32 + col];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const float av = As[2 * s + kh][i * 32 + col];
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
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

template <int MT, int TW>
int launch_fused_mt(const float *x, const float *packed, float *y, int B, int M, int K, int G, int nblk, hipStream_t s) {
    constexpr int TH = kBN / TW;
    const int tiles_x = (G + TW - 1) / TW, tiles_y = (G + TH - 1) / TH;
    const long nwork = (long)B * tiles_x * tiles_y * nblk;
    if (nwork > 0x7fffffffL) return gfn::fail(GFN_ERR_INVALID_ARG, "conv_block: too many tiles");
    // persistent workgroups: as many as the chip holds at once
    static int resident = 0;  // per instantiation
    if (!resident) {
        int dev = 0, cus = 0, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dwpw_fused_kernel<MT, TW>, 256, 0) != hipSuccess || cus <= 0 ||
            per_cu <= 0)
            return gfn::fail(GFN_ERR_LAUNCH, "conv_block: occupancy query failed");
        resident = cus * per_cu;
    }
    const unsigned grid = (unsigned)(nwork < resident ? nwork : resident);
    hipLaunchKernelGGL((dwpw_fused_kernel<MT, TW>), dim3(grid), dim3(256), 0, s, x, packed, y, M, K, G, tiles_x, tiles_y, nblk,
                       (unsigned)nwork);
    return gfn::check_launch("dwpw_fused_kernel");
}

template <int TW>
int launch_fused(int mt, const float *x, const float *packed, float *y, int B, int M, int K, int G, int nblk, hipStream_t s) {
    switch (mt) {
        case 1: return launch_fused_mt<1, TW>(x, packed, y, B, M, K, G, nblk, s);
        case 2: return launch_fused_mt<2, TW>(x, packed, y, B, M, K, G, nblk, s);
        case 3: return launch_fused_mt<3, TW>(x, packed, y, B, M, K, G, nblk, s);
        case 4: return launch_fused_mt<4, TW>(x, packed, y, B, M, K, G, nblk, s);
        case 5: return launch_fused_mt<5, TW>(x, packed, y, B, M, K, G, nblk, s);
        case 6: return launch_fused_mt<6, TW>(x, packed, y, B, M, K, G, nblk, s);
        default: return launch_fused_mt<7, TW>(x, packed, y, B, M, K, G, nblk, s);
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
    int nblk, mt;
    slab_shape(M, &nblk, &mt);
    const bool fused = variant != 1 && (G & 3) == 0;
    if (fused) {
        // tile width: full 128-byte rows where the map allows, narrower tiles for the 5*2^k grids
        if (G % 32 == 0 || G > 160) return launch_fused<32>(mt, x, packed, y, B, M, C, G, nblk, s);
        if (G % 16 == 0 || G > 64) return launch_fused<16>(mt, x, packed, y, B, M, C, G, nblk, s);
        return launch_fused<8>(mt, x, packed, y, B, M, C, G, nblk, s);
    }
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
#define GFN_PW(MT) hipLaunchKernelGGL((pw_gemm_kernel<MT>), grid, block, 0, s, packed, (const float *)t_scratch, y, M, C, N)
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
