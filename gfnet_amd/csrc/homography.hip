// homography.hip -- batched homography solve on gfx950.
//
// Replaces the solve at estimation.py:60-77 of the reference: cv2.findHomography(pos_a, pos_b,
// cv2.RANSAC, confidence=0.99999, ransacReprojThreshold=3) (OpenCV on the host after a
// device->host copy) and estimation.py:26-45 (convert_coordinates).  The pipeline is the published
// findHomography one, restated in oracle/homography_oracle.c, batched over image pairs and kept on
// the device:
//   hypothesis kernel : T minimal samples per pair (counter-based RNG), exact 4-point solve by
//                       Gaussian elimination with partial pivoting, one matrix row per lane
//                       (8 lanes per hypothesis, shuffles for pivot search / row swap)
//   score kernel      : one wave per hypothesis counts inliers (squared reprojection error
//                       <= thr^2) over the N correspondences, fp64, coalesced float4 reads
//   finish kernel     : one 1024-thread workgroup per pair: arg-max hypothesis (ties -> lowest
//                       index), inlier mask, normalised DLT on the inliers, <= 10 Levenberg-
//                       Marquardt steps.  Every 9x9 contraction (L^T L of the DLT, [J|r]^T [J|r] of
//                       LM) is accumulated on the matrix cores with v_mfma_f64_16x16x4_f64: each
//                       lane supplies ONE element L[k][i] as both the A and the B operand, 4 rows
//                       (2 correspondences) per instruction, so the reduction over N happens in
//                       the accumulator instead of 45 shuffle trees.  9x9 Jacobi eigen-solve and
//                       8x8 solves run on one wave with a matrix row per lane.
// The same kernels serve the one-shot weighted "grid-DLT" over a dense warp (weights = certainty).
// fp64 throughout; compiled with -ffp-contract=off so hypotheses, inlier counts and the chosen
// hypothesis are bit-identical to the oracle.
#include "common.h"

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

__device__ __forceinline__ uint32_t draw_index(uint64_t seed, uint32_t b, uint32_t t, uint32_t k, uint32_t a, uint32_t N) {
    uint64_t h = splitmix64(seed ^ splitmix64(((uint64_t)b << 32) | t));
    h = splitmix64(h + ((uint64_t)k << 8) + a);
    return (uint32_t)(h % N);
}

__device__ bool draw_sample(uint64_t seed, uint32_t b, uint32_t t, uint32_t N, uint32_t idx[4]) {
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        uint32_t a = 0;
        for (;;) {
            const uint32_t v = draw_index(seed, b, t, k, a, N);
            bool dup = false;
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) dup |= (q < k) & (idx[q] == v);
            if (!dup) { idx[k] = v; break; }
            if (++a >= 16) return false;
        }
    }
    return true;
}

// value of `v` in lane `srclane`.  UNIFORM: srclane is wave-uniform -> two v_readlane_b32 (a few
// cycles); otherwise a general shuffle (ds_bpermute, LDS crossbar latency).
template <bool UNIFORM>
__device__ __forceinline__ double lane_get(double v, int srclane) {
    if (UNIFORM) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
        return __hiloint2double(hi, lo);
    }
    return __shfl(v, srclane);
}

// Gaussian elimination with partial pivoting of an 8x8 system, one augmented row (9 doubles) per
// lane: lanes base..base+7 of the wave hold rows 0..7.  Same operation order as solve_aug() in the
// oracle.  Returns the solution component of this lane's row; ok is group-uniform.
// UNIFORM = the wave holds a single system (base is wave-uniform).
template <bool UNIFORM>
__device__ double ge_solve8(double (&M)[9], int row, int base, bool &ok) {
    ok = true;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        int piv = c;
        double best = fabs(lane_get<UNIFORM>(M[c], base + c));
#pragma unroll
        for (int r = c + 1; r < 8; ++r) {
            const double v = fabs(lane_get<UNIFORM>(M[c], base + r));
            if (v > best) { best = v; piv = r; }
        }
        if (!(best > 1e-300)) ok = false;
        if (UNIFORM) piv = __builtin_amdgcn_readfirstlane(piv);
        // swap rows c and piv (every lane takes part in the exchange)
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const double from_piv = lane_get<UNIFORM>(M[k], base + piv), from_c = lane_get<UNIFORM>(M[k], base + c);
            M[k] = (row == c) ? from_piv : ((row == piv) ? from_c : M[k]);
        }
        const double inv = 1.0 / lane_get<UNIFORM>(M[c], base + c);
        const double f = M[c] * inv;
#pragma unroll
        for (int k = c; k < 9; ++k) {
            const double prow = lane_get<UNIFORM>(M[k], base + c);
            if (row > c) M[k] = M[k] - f * prow;
        }
    }
    double s = M[8];
    double x = 0.0;
#pragma unroll
    for (int k = 7; k >= 0; --k) {
        // lane k finalises x_k = s / M[k][k]; every row above it (row < k) eliminates it
        const double xk = lane_get<UNIFORM>(s / M[k], base + k);
        if (row == k) x = xk;
        if (row < k) s = s - M[k] * xk;
    }
    return x;
}

__device__ __forceinline__ double reproj_err2(const double (&H)[9], const float4 p) {
    const double x = p.x, y = p.y, u = p.z, v = p.w;
    const double rw = 1.0 / (H[6] * x + H[7] * y + H[8]);  // one division; same sequence as the oracle
    const double dx = (H[0] * x + H[1] * y + H[2]) * rw - u;
    const double dy = (H[3] * x + H[4] * y + H[5]) * rw - v;
    return dx * dx + dy * dy;
}

// ---- kernel 1: hypotheses ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void hyp_kernel(const float *__restrict__ pts, double *__restrict__ hyp, int Bt, int N,
                                                  int T, uint64_t seed) {
    const int lane = threadIdx.x & 63;
    const int row = lane & 7, base = lane & ~7;
    const long gid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 3;  // hypothesis id
    const bool live = gid < (long)Bt * T;
    const int b = live ? (int)(gid / T) : 0, t = live ? (int)(gid - (long)b * T) : 0;
    uint32_t idx[4] = {0, 0, 0, 0};
    bool good = live && N >= 4 && draw_sample(seed, (uint32_t)b, (uint32_t)t, (uint32_t)N, idx);
    double M[9] = {1, 0, 0, 0, 0, 0, 0, 0, 0};
    if (good) {
        const int k = row >> 1;
        const uint32_t id = (k == 0) ? idx[0] : (k == 1) ? idx[1] : (k == 2) ? idx[2] : idx[3];
        const float4 p = reinterpret_cast<const float4 *>(pts)[(size_t)b * N + id];
        const double x = p.x, y = p.y, u = p.z, v = p.w;
        if (row & 1) {
            M[0] = 0; M[1] = 0; M[2] = 0; M[3] = x; M[4] = y; M[5] = 1; M[6] = -v * x; M[7] = -v * y; M[8] = v;
        } else {
            M[0] = x; M[1] = y; M[2] = 1; M[3] = 0; M[4] = 0; M[5] = 0; M[6] = -u * x; M[7] = -u * y; M[8] = u;
        }
    } else {
        M[0] = (row == 0); M[1] = (row == 1); M[2] = (row == 2); M[3] = (row == 3);
        M[4] = (row == 4); M[5] = (row == 5); M[6] = (row == 6); M[7] = (row == 7);
    }
    bool ok;
    double h = ge_solve8<false>(M, row, base, ok);
    good = good && ok;
    // every component must be finite
    const bool fin = isfinite(h);
    const unsigned long long allfin = __ballot(fin);
    good = good && (((allfin >> base) & 0xFFull) == 0xFFull);
    if (live) {
        double *o = hyp + (size_t)gid * 9;
        o[row] = good ? h : __builtin_nan("");
        if (row == 0) o[8] = good ? 1.0 : __builtin_nan("");
    }
}

// ---- kernel 2: inlier counts --------------------------------------------------------------------
__global__ __launch_bounds__(256) void score_kernel(const float *__restrict__ pts, const double *__restrict__ hyp,
                                                    int *__restrict__ counts, int Bt, int N, int T, double thr2) {
    const int lane = threadIdx.x & 63;
    const long gid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gid >= (long)Bt * T) return;
    const int b = (int)(gid / T);
    double H[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) H[k] = hyp[(size_t)gid * 9 + k];
    int c = 0;
    if (H[8] == 1.0) {  // NaN marks an invalid hypothesis
        const float4 *p = reinterpret_cast<const float4 *>(pts) + (size_t)b * N;
        for (int n = lane; n < N; n += 64) c += (reproj_err2(H, p[n]) <= thr2) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if (lane == 0) counts[gid] = c;
}

// ---- kernel 3: per-pair finish ------------------------------------------------------------------
constexpr int kFinThreads = 512;
constexpr int kFinWaves = kFinThreads / 64;

struct FinShared {
    double rows[kFinWaves][64][18];  // per wave: the two 9-element rows of 64 correspondences
    double gram[kFinWaves][81];
    double red[kFinWaves][12];
    double G[81];       // reduced gram
    double G2[81];
    double H[9];
    double h[9], hn[9];
    double stats[12];   // sw, cx, cy, cu, cv, sx, sy, su, sv
    unsigned long long key[kFinWaves];
    int cnt[kFinWaves];
    int flag;
    int best;
    int total;
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// sum K values per thread over the block (fixed order -> reproducible); result in sh.stats-like dst[0..K)
template <int K>
__device__ void block_sum(FinShared &sh, const double (&v)[K], double *dst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const double s = wave_sum(v[k]);
        if (lane == 0) sh.red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < K) {
        double s = 0;
        for (int w = 0; w < kFinWaves; ++w) s += sh.red[w][threadIdx.x];
        dst[threadIdx.x] = s;
    }
    __syncthreads();
}

// Accumulate sum over all correspondences of (rx rx^T + ry ry^T), rx, ry in R^9, on the f64 matrix
// core.  gen(n, rx, ry) fills the two rows of correspondence n (weight / mask already applied).
// Each lane generates the rows of ONE correspondence per step (64 per wave) and parks them in the
// wave's LDS slab; the MFMA operand of lane (i = lane&15, k = lane>>4) is then one LDS read:
// element i of row (k&1) of correspondence 2j + (k>>1) -- the same value serves as A and as B.
template <class Gen>
__device__ void gram_accumulate(FinShared &sh, int N, Gen gen, double *dst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, k = lane >> 4;
    f64x4 acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
    double(*rows)[18] = sh.rows[wave];
    for (int base = wave * 64; base < N; base += kFinWaves * 64) {
        const int n = base + lane;
        double rx[9], ry[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) { rx[q] = 0.0; ry[q] = 0.0; }
        if (n < N) gen(n, rx, ry);
#pragma unroll
        for (int q = 0; q < 9; ++q) { rows[lane][q] = rx[q]; rows[lane][9 + q] = ry[q]; }
        __builtin_amdgcn_wave_barrier();  // LDS ops of one wave execute in order; this only pins the compiler
        const int cnt = min(64, N - base);
        // two accumulation chains (even / odd steps) so that consecutive MFMAs do not wait on each other, operands of
        // four steps fetched ahead (-20 % on the LM loop)
        const double *col = &rows[k >> 1][(k & 1) * 9 + (i < 9 ? i : 0)];
        const int steps = (cnt + 1) >> 1;
        for (int j = 0; j < steps; j += 4) {
            double e[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) e[q] = (i < 9 && j + q < steps) ? col[(size_t)(2 * (j + q)) * 18] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(e[0], e[0], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(e[1], e[1], acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(e[2], e[2], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(e[3], e[3], acc2, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] += acc2[r];
    // D[row][col]: col = lane&15, row = (lane>>4) + 4*reg
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = k + 4 * r;
        if (row < 9 && i < 9) sh.gram[wave][row * 9 + i] = acc[r];
    }
    __syncthreads();
    if (threadIdx.x < 81) {
        double s = 0;
        for (int w = 0; w < kFinWaves; ++w) s += sh.gram[w][threadIdx.x];
        dst[threadIdx.x] = s;
    }
    __syncthreads();
}

__device__ __forceinline__ double rl(double v, int srclane) { return lane_get<true>(v, srclane); }  // srclane is a compile-time constant at every call

// cyclic Jacobi on wave 0: lane r (< 9) holds row r of A and of V.  Same rotation order and
// formulas as jacobi_eig() in the oracle.  On return lane r holds A[r][*] (diag = eigenvalues)
// and V[r][*] (columns = eigenvectors).
__device__ void jacobi9(double (&A)[9], double (&V)[9], int lane) {
#pragma unroll
    for (int j = 0; j < 9; ++j) V[j] = (lane == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0, diag = 0;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            if (lane < 9) {
                if (j == lane) diag += A[j] * A[j];
                if (j > lane) off += A[j] * A[j];
            }
        }
        off = wave_sum(off);
        diag = wave_sum(diag);
        if (off <= 1e-30 * diag || off == 0) break;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
#pragma unroll
            for (int q = p + 1; q < 9; ++q) {
                const double apq = rl(A[q], p), app = rl(A[p], p), aqq = rl(A[q], q);
                if (apq == 0) continue;  // wave-uniform
                const double theta = (aqq - app) / (2 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                const double c = 1 / sqrt(t * t + 1), s = t * c;
                // columns p,q of every row (lane-local)
                {
                    const double akp = A[p], akq = A[q];
                    A[p] = c * akp - s * akq;
                    A[q] = s * akp + c * akq;
                }
                // rows p,q: new row p = c*row p - s*row q (after the column update)
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const double apk = rl(A[k], p), aqk = rl(A[k], q);
                    if (lane == p) A[k] = c * apk - s * aqk;
                    if (lane == q) A[k] = s * apk + c * aqk;
                }
                {
                    const double vkp = V[p], vkq = V[q];
                    V[p] = c * vkp - s * vkq;
                    V[q] = s * vkp + c * vkq;
                }
            }
        }
    }
}

struct FinParams {
    const float *pts;       // (Bt,N,4)
    const float *weight;    // (Bt,N) or null (dlt mode)
    const double *hyp;      // (Bt,T,9)   (ransac mode)
    const int *counts;      // (Bt,T)
    double *H;              // (Bt,9)
    int *ninl;              // (Bt)
    int *best_t;            // (Bt)
    unsigned char *mask;    // (Bt,N) or null
    int N, T;
    double thr2;
    int lm_iters;
    int stage;              // 0 full, 1 stop after RANSAC, 2 stop after DLT
    int mode;               // 0 = ransac pipeline, 1 = one-shot weighted DLT
};

__global__ __launch_bounds__(kFinThreads) void finish_kernel(FinParams P) {
    __shared__ FinShared sh;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = P.N;
    const float4 *pts = reinterpret_cast<const float4 *>(P.pts) + (size_t)b * N;
    const float *wgt = P.weight ? P.weight + (size_t)b * N : nullptr;
    unsigned char *mask = P.mask ? P.mask + (size_t)b * N : nullptr;

    int cnt = 0;
    if (P.mode == 0) {
        // ---- arg-max over hypotheses: key = (count << 32) | ~t  -> max count, lowest t ----------
        unsigned long long key = 0;
        for (int t = tid; t < P.T; t += kFinThreads) {
            const unsigned long long k = ((unsigned long long)(unsigned)P.counts[(size_t)b * P.T + t] << 32) |
                                         (unsigned long long)(0xFFFFFFFFu - (unsigned)t);
            key = k > key ? k : key;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(key, o);
            key = other > key ? other : key;
        }
        if (lane == 0) sh.key[wave] = key;
        __syncthreads();
        if (tid == 0) {
            unsigned long long k = 0;
            for (int w = 0; w < kFinWaves; ++w) k = sh.key[w] > k ? sh.key[w] : k;
            const int c = (int)(k >> 32);
            sh.best = c > 0 ? (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFu)) : -1;
        }
        __syncthreads();
        const int best = sh.best;
        if (tid < 9) sh.H[tid] = best >= 0 ? P.hyp[((size_t)b * P.T + best) * 9 + tid] : (tid == 8 ? 1.0 : 0.0);
        __syncthreads();
        // ---- inlier mask of the best hypothesis ------------------------------------------------
        double Hb[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) Hb[k] = sh.H[k];
        int c = 0;
        for (int n = tid; n < N; n += kFinThreads) {
            const bool in = best >= 0 && reproj_err2(Hb, pts[n]) <= P.thr2;
            if (mask) mask[n] = in ? 1 : 0;
            c += in ? 1 : 0;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        if (lane == 0) sh.cnt[wave] = c;
        __syncthreads();
        if (tid == 0) {
            int s = 0;
            for (int w = 0; w < kFinWaves; ++w) s += sh.cnt[w];
            sh.total = s;
        }
        __syncthreads();
        cnt = sh.total;
        if (best < 0 || cnt < 4) {  // estimation.py:74-76: failure -> diag(0,0,1)
            if (tid < 9) P.H[(size_t)b * 9 + tid] = tid == 8 ? 1.0 : 0.0;
            if (tid == 0) { P.ninl[b] = cnt; P.best_t[b] = -1; }
            return;
        }
        if (tid == 0) { P.ninl[b] = cnt; P.best_t[b] = best; }
        if (P.stage == 1 || cnt <= 4) {
            if (tid < 9) P.H[(size_t)b * 9 + tid] = sh.H[tid];
            return;
        }
    }

    // per-point weight (mask in ransac mode, certainty in dlt mode)
    auto point_w = [&](int n, const double (&Hb)[9]) -> double {
        if (P.mode == 0) return reproj_err2(Hb, pts[n]) <= P.thr2 ? 1.0 : 0.0;
        return wgt ? (double)wgt[n] : 1.0;
    };
    double Hb[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) Hb[k] = (P.mode == 0) ? sh.H[k] : 0.0;

    // ---- normalised DLT: centroid and mean absolute deviation (OpenCV runKernel) -----------------
    {
        double v[5] = {0, 0, 0, 0, 0};
        for (int n = tid; n < N; n += kFinThreads) {
            const double w = point_w(n, Hb);
            const float4 p = pts[n];
            v[0] += w; v[1] += w * p.x; v[2] += w * p.y; v[3] += w * p.z; v[4] += w * p.w;
        }
        block_sum<5>(sh, v, sh.stats);
    }
    const double sw = sh.stats[0];
    bool dlt_ok = sw > 0;
    const double cx = sh.stats[1] / sw, cy = sh.stats[2] / sw, cu = sh.stats[3] / sw, cv = sh.stats[4] / sw;
    __syncthreads();
    {
        double v[4] = {0, 0, 0, 0};
        for (int n = tid; n < N; n += kFinThreads) {
            const double w = point_w(n, Hb);
            const float4 p = pts[n];
            v[0] += w * fabs(p.x - cx); v[1] += w * fabs(p.y - cy); v[2] += w * fabs(p.z - cu); v[3] += w * fabs(p.w - cv);
        }
        block_sum<4>(sh, v, sh.stats + 5);
    }
    dlt_ok = dlt_ok && sh.stats[5] > 1e-300 && sh.stats[6] > 1e-300 && sh.stats[7] > 1e-300 && sh.stats[8] > 1e-300;
    const double sx = sw / sh.stats[5], sy = sw / sh.stats[6], su = sw / sh.stats[7], sv = sw / sh.stats[8];
    if (dlt_ok) {
        gram_accumulate(sh, N, [&](int n, double (&rx)[9], double (&ry)[9]) {
            const double sw_ = sqrt(point_w(n, Hb));
            const float4 p = pts[n];
            const double X = (p.x - cx) * sx * sw_, Y = (p.y - cy) * sy * sw_, x = (p.z - cu) * su, y = (p.w - cv) * sv;
            rx[0] = X; rx[1] = Y; rx[2] = sw_; rx[6] = -x * X; rx[7] = -x * Y; rx[8] = -x * sw_;
            ry[3] = X; ry[4] = Y; ry[5] = sw_; ry[6] = -y * X; ry[7] = -y * Y; ry[8] = -y * sw_;
        }, sh.G);
        // eigenvector of the smallest eigenvalue (wave 0, one matrix row per lane)
        if (wave == 0) {
            double A[9], V[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) A[j] = (lane < 9) ? sh.G[lane * 9 + j] : 0.0;
            jacobi9(A, V, lane);
            // eigenvalue r sits at A[r] of lane r
            double ev = 0;
#pragma unroll
            for (int j = 0; j < 9; ++j) if (lane == j) ev = A[j];
            int kmin = 0;
            double best = rl(ev, 0);
#pragma unroll
            for (int k = 1; k < 9; ++k) {
                const double e = rl(ev, k);
                if (e < best) { best = e; kmin = k; }
            }
            double hk = 0;  // h[lane] = V[lane][kmin]
#pragma unroll
            for (int j = 0; j < 9; ++j) if (j == kmin) hk = V[j];
            if (lane < 9) sh.h[lane] = hk;
        }
        __syncthreads();
        if (tid == 0) {
            double h[9], A[9], Hn[9];
            for (int k = 0; k < 9; ++k) h[k] = sh.h[k];
            for (int c = 0; c < 3; ++c) {
                A[0 + c] = h[0 + c] / su + cu * h[6 + c];
                A[3 + c] = h[3 + c] / sv + cv * h[6 + c];
                A[6 + c] = h[6 + c];
            }
            for (int r = 0; r < 3; ++r) {
                Hn[3 * r + 0] = A[3 * r + 0] * sx;
                Hn[3 * r + 1] = A[3 * r + 1] * sy;
                Hn[3 * r + 2] = A[3 * r + 2] - A[3 * r + 0] * cx * sx - A[3 * r + 1] * cy * sy;
            }
            bool good = fabs(Hn[8]) > 1e-300;
            const double inv = 1.0 / Hn[8];
            for (int k = 0; k < 9; ++k) { Hn[k] *= inv; good = good && isfinite(Hn[k]); }
            sh.flag = good ? 1 : 0;
            if (good) for (int k = 0; k < 9; ++k) sh.H[k] = Hn[k];
        }
        __syncthreads();
        dlt_ok = sh.flag != 0;
    }
    if (P.mode == 1) {
        if (tid < 9) P.H[(size_t)b * 9 + tid] = dlt_ok ? sh.H[tid] : (tid == 8 ? 1.0 : 0.0);
        if (tid == 0) P.ninl[b] = dlt_ok ? 1 : 0;
        return;
    }
    if (P.stage == 2) {
        if (tid < 9) P.H[(size_t)b * 9 + tid] = sh.H[tid];
        return;
    }

    // ---- Levenberg-Marquardt on the inliers (mask fixed by the RANSAC hypothesis Hb) --------------
    auto lm_gram = [&](const double *hsrc, double *dst) {
        double h[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) h[k] = hsrc[k];
        gram_accumulate(sh, N, [&](int n, double (&rx)[9], double (&ry)[9]) {
            const double w = point_w(n, Hb);  // 0/1 here
            const float4 p = pts[n];
            const double X = p.x, Y = p.y, u = p.z, v = p.w;
            const double ww = w != 0.0 ? 1.0 / (h[6] * X + h[7] * Y + 1.0) : 0.0;  // masked points contribute exact zeros
            const double xi = (h[0] * X + h[1] * Y + h[2]) * ww, yi = (h[3] * X + h[4] * Y + h[5]) * ww;
            // xi, yi carry the weight; for w = 1 these are the reference's Jacobian rows and residuals
            rx[0] = X * ww; rx[1] = Y * ww; rx[2] = ww; rx[6] = -X * ww * xi; rx[7] = -Y * ww * xi; rx[8] = xi - w * u;
            ry[3] = X * ww; ry[4] = Y * ww; ry[5] = ww; ry[6] = -X * ww * yi; ry[7] = -Y * ww * yi; ry[8] = yi - w * v;
        }, dst);
    };
    if (tid < 9) sh.h[tid] = sh.H[tid] / sh.H[8];
    __syncthreads();
    lm_gram(sh.h, sh.G);
    double S = sh.G[80], lambda = 1e-3;
    for (int it = 0; it < P.lm_iters; ++it) {
        // solve (G8 + lambda diag) delta = -g on wave 0, one row per lane
        if (wave == 0) {
            double M[9];
            const int row = lane & 7;
#pragma unroll
            for (int j = 0; j < 8; ++j) M[j] = sh.G[row * 9 + j] + ((j == row) ? lambda * sh.G[row * 9 + row] : 0.0);
            M[8] = -sh.G[row * 9 + 8];
            bool ok;
            const double d = ge_solve8<true>(M, row, 0, ok);
            if (lane < 8) sh.hn[lane] = sh.h[lane] + d;
            if (lane == 0) sh.hn[8] = 1.0;
            // step / parameter norms for the stopping rule
            double dn = (lane < 8) ? d * d : 0.0, hn2 = (lane < 8) ? sh.h[lane] * sh.h[lane] : 0.0;
            dn = wave_sum(dn);
            hn2 = wave_sum(hn2);
            if (lane == 0) { sh.flag = ok ? 1 : 0; sh.stats[10] = dn; sh.stats[11] = hn2; }
        }
        __syncthreads();
        if (!sh.flag) { lambda *= 10; __syncthreads(); continue; }
        lm_gram(sh.hn, sh.G2);
        const double S2 = sh.G2[80];
        const bool accept = S2 < S;
        const bool stop = accept && sh.stats[10] <= 1e-24 * (sh.stats[11] + 1e-24);
        __syncthreads();
        if (accept) {
            if (tid < 9) sh.h[tid] = sh.hn[tid];
            if (tid < 81) sh.G[tid] = sh.G2[tid];
            S = S2;
            lambda = lambda > 1e-11 ? lambda / 10 : 1e-12;
        } else {
            lambda *= 10;
        }
        __syncthreads();
        if (stop) break;
    }
    if (tid < 9) P.H[(size_t)b * 9 + tid] = sh.h[tid];
}

__global__ __launch_bounds__(256) void convert_matches_kernel(const float *__restrict__ m, float *__restrict__ pts, long n,
                                                              float wA, float hA, float wB, float hB) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float4 v = reinterpret_cast<const float4 *>(m)[idx];
    float4 o;  // estimation.py:26-45 in float32: (w-1) * (x+1) / 2
    o.x = ((wA - 1.f) * (v.x + 1.f)) / 2.f;
    o.y = ((hA - 1.f) * (v.y + 1.f)) / 2.f;
    o.z = ((wB - 1.f) * (v.z + 1.f)) / 2.f;
    o.w = ((hB - 1.f) * (v.w + 1.f)) / 2.f;
    reinterpret_cast<float4 *>(pts)[idx] = o;
}

}  // namespace

GFN_EXPORT int gfn_convert_matches(const float *matches, float *pts, int64_t n, float wA, float hA, float wB, float hB,
                                   gfn_stream_t stream) {
    if (!matches || !pts || n < 0) return gfn::fail(GFN_ERR_INVALID_ARG, "convert_matches: bad argument");
    if (((uintptr_t)matches | (uintptr_t)pts) & 15) return gfn::fail(GFN_ERR_INVALID_ARG, "convert_matches: need 16-byte aligned rows");
    if (n == 0) return GFN_OK;
    hipLaunchKernelGGL(convert_matches_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, matches,
                       pts, (long)n, wA, hA, wB, hB);
    return gfn::check_launch("convert_matches_kernel");
}

GFN_EXPORT int64_t gfn_homography_scratch_bytes(int Bt, int iters) {
    // hypotheses (9 doubles) + counts (int), rounded up
    return (int64_t)Bt * iters * (9 * 8 + 4) + 256;
}

GFN_EXPORT int gfn_homography_ransac(const float *pts, int Bt, int N, double thresh, int iters, uint64_t seed, int lm_iters,
                                     int stage, double *H, int *ninl, int *best_t, unsigned char *mask, void *scratch,
                                     int64_t scratch_bytes, gfn_stream_t stream) {
    if (!pts || !H || !ninl || !best_t) return gfn::fail(GFN_ERR_INVALID_ARG, "homography_ransac: null pointer");
    if (Bt < 0 || N < 0 || iters <= 0 || !(thresh > 0) || lm_iters < 0 || stage < 0 || stage > 2)
        return gfn::fail(GFN_ERR_INVALID_ARG, "homography_ransac: bad argument");
    if ((uintptr_t)pts & 15) return gfn::fail(GFN_ERR_INVALID_ARG, "homography_ransac: pts must be 16-byte aligned");
    if (!scratch || scratch_bytes < gfn_homography_scratch_bytes(Bt, iters))
        return gfn::fail(GFN_ERR_SCRATCH, "homography_ransac: scratch too small (%lld < %lld bytes)", (long long)scratch_bytes,
                         (long long)gfn_homography_scratch_bytes(Bt, iters));
    if (Bt == 0) return GFN_OK;
    hipStream_t s = (hipStream_t)stream;
    double *hyp = reinterpret_cast<double *>(scratch);
    int *counts = reinterpret_cast<int *>(hyp + (size_t)Bt * iters * 9);
    const long nh = (long)Bt * iters;
    hipLaunchKernelGGL(hyp_kernel, dim3((unsigned)((nh * 8 + 255) / 256)), dim3(256), 0, s, pts, hyp, Bt, N, iters, seed);
    if (int e = gfn::check_launch("hyp_kernel")) return e;
    hipLaunchKernelGGL(score_kernel, dim3((unsigned)((nh + 3) / 4)), dim3(256), 0, s, pts, hyp, counts, Bt, N, iters,
                       thresh * thresh);
    if (int e = gfn::check_launch("score_kernel")) return e;
    FinParams P;
    P.pts = pts; P.weight = nullptr; P.hyp = hyp; P.counts = counts; P.H = H; P.ninl = ninl; P.best_t = best_t; P.mask = mask;
    P.N = N; P.T = iters; P.thr2 = thresh * thresh; P.lm_iters = lm_iters; P.stage = stage; P.mode = 0;
    hipLaunchKernelGGL(finish_kernel, dim3(Bt), dim3(kFinThreads), 0, s, P);
    return gfn::check_launch("finish_kernel");
}

GFN_EXPORT int gfn_homography_dlt(const float *pts, const float *weight, int Bt, int N, double *H, int *ok,
                                  gfn_stream_t stream) {
    if (!pts || !H || !ok || Bt < 0 || N < 0) return gfn::fail(GFN_ERR_INVALID_ARG, "homography_dlt: bad argument");
    if ((uintptr_t)pts & 15) return gfn::fail(GFN_ERR_INVALID_ARG, "homography_dlt: pts must be 16-byte aligned");
    if (Bt == 0) return GFN_OK;
    FinParams P;
    P.pts = pts; P.weight = weight; P.hyp = nullptr; P.counts = nullptr; P.H = H; P.ninl = ok; P.best_t = nullptr; P.mask = nullptr;
    P.N = N; P.T = 0; P.thr2 = 0; P.lm_iters = 0; P.stage = 0; P.mode = 1;
    hipLaunchKernelGGL(finish_kernel, dim3(Bt), dim3(kFinThreads), 0, (hipStream_t)stream, P);
    return gfn::check_launch("finish_kernel(dlt)");
}
