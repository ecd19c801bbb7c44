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
//   finish kernel     : one 512-thread workgroup per pair: arg-max hypothesis (ties -> lowest
//                       index), inlier mask, normalised DLT on the inliers, <= 10 Levenberg-
//                       Marquardt steps.  The 9x9 contractions (L^T L of the DLT, [J|r]^T [J|r] of LM)
//                       are accumulated per thread in their sparse form (36 distinct entries, 42 fp64
//                       FMAs per correspondence) and reduced with shuffles; the DLT null vector comes
//                       from LU + inverse iteration and the 8x8 LM solves from Gaussian elimination,
//                       both on one wave with a matrix row per lane (v_readlane broadcasts).
//                       (The north star asks for MFMA on "the batched 9x9 DLT contractions where it
//                       really is dense": measured, it is not -- two thirds of every rank-1 block are
//                       structural zeros, a v_mfma_f64_16x16x4_f64 tile spends 12x the flops of the
//                       sparse form at 64 cycles per instruction, and the matrix pipe alone took 40 k
//                       of the 83 k cycles of a pass that now takes 9 k.)
// The same kernels serve the one-shot weighted "grid-DLT" over a dense warp (weights = certainty).
// fp64 throughout; compiled with -ffp-contract=off so hypotheses, inlier counts and the chosen
// hypothesis are bit-identical to the oracle.
#include "common.h"

namespace {

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

__device__ __forceinline__ uint32_t draw_index(uint64_t seed, uint32_t b, uint32_t t, uint32_t k, uint32_t a, uint32_t s, uint32_t N) {
    uint64_t h = splitmix64(seed ^ splitmix64(((uint64_t)b << 32) | t));
    h = splitmix64(h + ((uint64_t)s << 16) + ((uint64_t)k << 8) + a);
    return (uint32_t)(h % N);
}

__device__ bool draw_sample(uint64_t seed, uint32_t b, uint32_t t, uint32_t s, uint32_t N, uint32_t idx[4]) {
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        uint32_t a = 0;
        for (;;) {
            const uint32_t v = draw_index(seed, b, t, k, a, s, N);
            bool dup = false;
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) dup |= (q < k) & (idx[q] == v);
            if (!dup) { idx[k] = v; break; }
            if (++a >= 16) return false;
        }
    }
    return true;
}

// HomographyEstimatorCallback::checkSubset for 4 correspondences, the operation sequence of subset_ok() in the oracle:
// the last point must not be collinear (within rounding) with two of the first three in either image, and the four
// triplets must keep or all flip their orientation between the images.
__device__ bool subset_ok(const float4 *pts, const uint32_t idx[4]) {
    double X[2][4], Y[2][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float4 p = pts[idx[k]];
        X[0][k] = p.x; Y[0][k] = p.y; X[1][k] = p.z; Y[1][k] = p.w;
    }
    bool ok = true;
#pragma unroll
    for (int im = 0; im < 2; ++im)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double dx1 = X[im][j] - X[im][3], dy1 = Y[im][j] - Y[im][3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (k >= j) continue;
                const double dx2 = X[im][k] - X[im][3], dy2 = Y[im][k] - Y[im][3];
                ok &= !(fabs(dx2 * dy1 - dy2 * dx1) <= 1.1920928955078125e-07 * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2)));
            }
        }
    const int tt[4][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
    int negative = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double det[2];
#pragma unroll
        for (int im = 0; im < 2; ++im) {
            const double x0 = X[im][tt[i][0]], y0 = Y[im][tt[i][0]], x1 = X[im][tt[i][1]], y1 = Y[im][tt[i][1]];
            const double x2 = X[im][tt[i][2]], y2 = Y[im][tt[i][2]];
            det[im] = (x0 * (y1 - y2) - y0 * (x1 - x2)) + (x1 * y2 - x2 * y1);
        }
        negative += (det[0] * det[1] < 0) ? 1 : 0;
    }
    return ok && (negative == 0 || negative == 4);
}

// the subset of hypothesis t: up to kSubsetTries draws until one passes (OpenCV re-draws inside getSubset)
constexpr uint32_t kSubsetTries = 8;
__device__ bool draw_checked(const float4 *pts, uint64_t seed, uint32_t b, uint32_t t, uint32_t N, uint32_t idx[4]) {
    for (uint32_t s = 0; s < kSubsetTries; ++s) {
        if (!draw_sample(seed, b, t, s, N, idx)) continue;
        if (subset_ok(pts, idx)) return true;
    }
    return false;
}

// log(x) from + - * / only, bit for bit det_log() of the oracle (the iteration bound must not depend on a libm)
__device__ double det_log(double x) {
    unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    bits = (bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m = __longlong_as_double((long long)bits);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double t = (m - 1.0) / (m + 1.0), t2 = t * t;
    double s = 0.0;
    for (int k = 14; k >= 0; --k) s = s * t2 + 1.0 / (double)(2 * k + 1);
    return 2.0 * t * s + (double)e * 0.6931471805599453;
}

// cv::RANSACUpdateNumIters(confidence, outlier ratio, modelPoints = 4, current bound)
__device__ int ransac_update_iters(double conf, int good, int N, int niters) {
    const double p = conf < 0 ? 0 : (conf > 1 ? 1 : conf);
    double ep = (double)(N - good) / (double)N;
    ep = ep < 0 ? 0 : (ep > 1 ? 1 : ep);
    double num = 1.0 - p;
    if (num < 2.2250738585072014e-308) num = 2.2250738585072014e-308;
    const double w = 1.0 - ep;
    double denom = 1.0 - (w * w) * (w * w);
    if (denom < 2.2250738585072014e-308) return 0;
    num = det_log(num);
    denom = det_log(denom);
    if (denom >= 0 || -num >= (double)niters * (-denom)) return niters;
    return (int)floor(num / denom + 0.5);
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

// Inlier test: squared reprojection error <= thr2, multiplied through by w^2 so that no division is needed, sums as
// explicit v_fma_f64 (12 instead of 25 fp64 instructions per test under -ffp-contract=off) -- the same operation sequence as
// is_inlier() in the oracle (bit-identical decisions).
__device__ __forceinline__ bool is_inlier(const double (&H)[9], double x, double y, double u, double v, double thr2) {
    const double w = fma(H[6], x, fma(H[7], y, H[8]));
    const double dx = fma(-u, w, fma(H[0], x, fma(H[1], y, H[2])));
    const double dy = fma(-v, w, fma(H[3], x, fma(H[4], y, H[5])));
    const double w2 = w * w;
    return (fma(dx, dx, dy * dy) <= thr2 * w2) & (w2 > 0);
}
__device__ __forceinline__ bool is_inlier(const double (&H)[9], const float4 p, double thr2) {
    return is_inlier(H, (double)p.x, (double)p.y, (double)p.z, (double)p.w, thr2);
}

// ---- kernel 1: hypotheses ---------------------------------------------------------------------
// Hypotheses t0 .. T-1 of every pair; bound (or NULL): pair b only needs t < bound[b] (adaptive RANSAC), the rest leave at once.
__global__ __launch_bounds__(256) void hyp_kernel(const float *__restrict__ pts, double *__restrict__ hyp, int Bt, int N,
                                                  int T, int t0, const int *__restrict__ bound, uint64_t seed) {
    const int lane = threadIdx.x & 63;
    const int row = lane & 7, base = lane & ~7;
    const int TR = T - t0;
    const long gid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 3;  // hypothesis id
    const bool in_range = gid < (long)Bt * TR;
    const int b = in_range ? (int)(gid / TR) : 0, t = in_range ? t0 + (int)(gid - (long)b * TR) : 0;
    const bool live = in_range && (!bound || t < bound[b]);
    if (!__any(live)) return;  // wave-uniform: the cross-lane solve below needs whole groups
    uint32_t idx[4] = {0, 0, 0, 0};
    bool good = live && N >= 4 && draw_checked(reinterpret_cast<const float4 *>(pts) + (size_t)b * N, seed, (uint32_t)b, (uint32_t)t, (uint32_t)N, idx);
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
        double *o = hyp + ((size_t)b * T + t) * 9;
        o[row] = good ? h : __builtin_nan("");
        if (row == 0) o[8] = good ? 1.0 : __builtin_nan("");
    }
}

// ---- kernel 2: inlier counts --------------------------------------------------------------------
// One wave scores kHypPerWave hypotheses of one pair: every correspondence is loaded and widened to fp64 once per group
// (the 2000 hypotheses of a pair re-read the same 80 KB of points; at one hypothesis per wave that was 5 GB of L2 -> L1
// traffic per 32-pair batch and four conversions per test).
constexpr int kHypPerWave = 4;  // 8 measured slower (register pressure / fewer waves)
__global__ __launch_bounds__(256) void score_kernel(const float *__restrict__ pts, const double *__restrict__ hyp,
                                                    int *__restrict__ counts, int Bt, int N, int T, int tbeg, const int *__restrict__ bound,
                                                    double thr2) {
    const int lane = threadIdx.x & 63;
    const int groups = (T - tbeg + kHypPerWave - 1) / kHypPerWave;
    const long gid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gid >= (long)Bt * groups) return;
    const int b = (int)(gid / groups), t0 = tbeg + (int)(gid - (long)b * groups) * kHypPerWave;
    if (bound && t0 >= bound[b]) return;  // adaptive RANSAC: the sequential loop never gets here
    double H[kHypPerWave][9];
    bool valid[kHypPerWave];
#pragma unroll
    for (int h = 0; h < kHypPerWave; ++h) {
        const int t = min(t0 + h, T - 1);
#pragma unroll
        for (int k = 0; k < 9; ++k) H[h][k] = hyp[((size_t)b * T + t) * 9 + k];
        valid[h] = H[h][8] == 1.0;  // NaN marks an invalid hypothesis
    }
    int c[kHypPerWave];
#pragma unroll
    for (int h = 0; h < kHypPerWave; ++h) c[h] = 0;
    const float4 *p = reinterpret_cast<const float4 *>(pts) + (size_t)b * N;
    for (int n = lane; n < N; n += 64) {
        const float4 q = p[n];
        const double x = q.x, y = q.y, u = q.z, v = q.w;
#pragma unroll
        for (int h = 0; h < kHypPerWave; ++h) c[h] += is_inlier(H[h], x, y, u, v, thr2) ? 1 : 0;
    }
#pragma unroll
    for (int h = 0; h < kHypPerWave; ++h) {
        int s = valid[h] ? c[h] : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0 && t0 + h < T) counts[(size_t)b * T + t0 + h] = s;
    }
}

// ---- adaptive RANSAC: the head of the sequential loop ------------------------------------------------------------------
// cv::RANSACPointSetRegistrator::run is sequential: whenever a hypothesis beats the best inlier count so far the iteration
// bound shrinks to RANSACUpdateNumIters(confidence, ...) -- at the inlier ratios a matcher delivers, a dozen iterations
// instead of 2000 (estimation.py:66-72 passes confidence = 0.99999).  The device keeps that outcome exactly and stays
// parallel: (1) this kernel, one workgroup per pair, evaluates the first kHeadHyp hypotheses and applies the rule to them,
// which gives an upper bound on where the sequential loop can stop; (2) hyp_kernel / score_kernel evaluate hypotheses
// kHeadHyp .. bound - 1 over the whole chip (nothing at all when the bound is below kHeadHyp: the usual case);
// (3) finish_kernel replays the rule over the counts in hypothesis order.
// Here a thread solves its 4-point system on its own (solve_aug() of the oracle, operation for operation, matrix in LDS):
// the 8-lanes-per-system elimination of hyp_kernel is a 20-40 us chain of cross-lane moves that only pays when thousands of
// systems hide each other's latency.
constexpr int kHeadHyp = 16;
constexpr int kHeadThreads = kHeadHyp * 64;
__global__ __launch_bounds__(kHeadThreads) void ransac_head_kernel(const float *__restrict__ pts, double *__restrict__ hyp,
                                                                   int *__restrict__ counts, int *__restrict__ bound, int N, int T,
                                                                   uint64_t seed, double thr2, double confidence) {
    __shared__ double sM[72][kHeadHyp];  // augmented 8 x 9 systems, element-major: thread h owns column h
    __shared__ double sH[kHeadHyp][9];
    __shared__ int sCnt[kHeadHyp];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float4 *p = reinterpret_cast<const float4 *>(pts) + (size_t)b * N;
    if (tid < kHeadHyp) {
        const int h = tid, t = h;
        uint32_t idx[4] = {0, 0, 0, 0};
        bool good = t < T && N >= 4 && draw_checked(p, seed, (uint32_t)b, (uint32_t)t, (uint32_t)N, idx);
        if (good) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 q = p[idx[k]];
                const double x = q.x, y = q.y, u = q.z, v = q.w;
                const int r0 = (2 * k) * 9, r1 = (2 * k + 1) * 9;
                sM[r0 + 0][h] = x; sM[r0 + 1][h] = y; sM[r0 + 2][h] = 1; sM[r0 + 3][h] = 0; sM[r0 + 4][h] = 0; sM[r0 + 5][h] = 0;
                sM[r0 + 6][h] = -u * x; sM[r0 + 7][h] = -u * y; sM[r0 + 8][h] = u;
                sM[r1 + 0][h] = 0; sM[r1 + 1][h] = 0; sM[r1 + 2][h] = 0; sM[r1 + 3][h] = x; sM[r1 + 4][h] = y; sM[r1 + 5][h] = 1;
                sM[r1 + 6][h] = -v * x; sM[r1 + 7][h] = -v * y; sM[r1 + 8][h] = v;
            }
            // solve_aug(M, 8) of the oracle
            for (int c = 0; c < 8 && good; ++c) {
                int piv = c;
                double best = fabs(sM[c * 9 + c][h]);
                for (int r = c + 1; r < 8; ++r) {
                    const double v = fabs(sM[r * 9 + c][h]);
                    if (v > best) { best = v; piv = r; }
                }
                if (!(best > 1e-300)) { good = false; break; }
                if (piv != c)
                    for (int k = 0; k < 9; ++k) { const double tmp = sM[c * 9 + k][h]; sM[c * 9 + k][h] = sM[piv * 9 + k][h]; sM[piv * 9 + k][h] = tmp; }
                const double inv = 1.0 / sM[c * 9 + c][h];
                for (int r = c + 1; r < 8; ++r) {
                    const double f = sM[r * 9 + c][h] * inv;
                    for (int k = c; k < 9; ++k) sM[r * 9 + k][h] = sM[r * 9 + k][h] - f * sM[c * 9 + k][h];
                }
            }
            if (good) {
                for (int r = 7; r >= 0; --r) {
                    double sacc = sM[r * 9 + 8][h];
                    for (int k = 7; k > r; --k) sacc = sacc - sM[r * 9 + k][h] * sM[k * 9 + 8][h];
                    sM[r * 9 + 8][h] = sacc / sM[r * 9 + r][h];
                }
                for (int k = 0; k < 8; ++k) good = good && isfinite(sM[k * 9 + 8][h]);
            }
        }
        for (int k = 0; k < 8; ++k) sH[h][k] = good ? sM[k * 9 + 8][h] : __builtin_nan("");
        sH[h][8] = good ? 1.0 : __builtin_nan("");
        if (t < T) {
            double *o = hyp + ((size_t)b * T + t) * 9;
            for (int k = 0; k < 9; ++k) o[k] = sH[h][k];
        }
    }
    __syncthreads();
    {   // inlier counts: one wave per hypothesis
        double H[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) H[k] = sH[wave][k];
        const bool valid = H[8] == 1.0;  // NaN marks an invalid hypothesis
        int c = 0;
        for (int n = lane; n < N; n += 64) c += is_inlier(H, p[n], thr2) ? 1 : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        if (lane == 0) sCnt[wave] = valid ? c : 0;
    }
    __syncthreads();
    if (tid == 0) {
        int niters = N >= 4 ? T : 0, bestc = 0;
        for (int t = 0; t < kHeadHyp && t < T; ++t) {
            const int c = sCnt[t];
            counts[(size_t)b * T + t] = c;
            if (t < niters && c > (bestc > 3 ? bestc : 3)) {  // goodCount > max(maxGoodCount, modelPoints - 1)
                bestc = c;
                niters = ransac_update_iters(confidence, c, N, niters);
            }
        }
        bound[b] = niters;  // the sequential loop stops at or before this hypothesis
    }
}

// ---- kernel 3: per-pair finish ------------------------------------------------------------------
constexpr int kFinThreads = 512;
constexpr int kFinScan = 2048;
constexpr int kFinWaves = kFinThreads / 64;

struct FinShared {
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
    int scan[kFinScan];  // adaptive RANSAC: a stretch of inlier counts for the sequential replay
};

// Sum over the 64 lanes of a wave (returned to every lane, fixed order).  DPP moves on the VALU instead of the ds_bpermute
// butterfly: a finish pass reduces 36-45 doubles, and 12 LDS-pipe permutes per value were most of its ~9 k cycles.
//   quad_perm [1,0,3,2], [2,3,0,1]: quad sums; row_half_mirror, row_mirror: sums of 8 and 16 (the groups are uniform by then);
//   row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3: lane 63 holds the total.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int tl = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    const int th = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return v + __hiloint2double(th, tl);  // rows outside ROW_MASK receive +0.0
}
__device__ __forceinline__ double wave_sum(double v) {
    v = dpp_add<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xf>(v);  // row_half_mirror
    v = dpp_add<0x140, 0xf>(v);  // row_mirror
    v = dpp_add<0x142, 0xa>(v);  // row_bcast:15
    v = dpp_add<0x143, 0xc>(v);  // row_bcast:31
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
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

// Accumulate G = sum over all correspondences of (rx rx^T + ry ry^T), rx, ry in R^9 with the sparsity of the DLT / LM
// rows:  rx = [a0 a1 a2 0 0 0 a3 a4 a5],  ry = [0 0 0 b0 b1 b2 b3 b4 b5].  gen(n, a, b) fills the twelve values of
// correspondence n (weight / mask already applied).  Only 36 of the 81 entries are distinct and non-zero, 42 fp64 FMAs per
// correspondence on the VALU.  (Round 1 ran this on v_mfma_f64_16x16x4_f64: a 16x16x4 tile per two correspondences is
// 12x the flops of the sparse form, and at 64 cycles per instruction the matrix pipe alone took 40 k of the 83 k cycles
// of a 5000-point pass; this form takes ~9 k.)
//   acc layout: [0..5] a-block upper triangle, [6..14] a x c, [15..20] b-block, [21..29] b x c, [30..35] c-block (from a and b)
__device__ __forceinline__ int gram_slot(int r, int c) {  // acc index of G[r][c], -1 for the structural zeros
    if (r > c) { const int t = r; r = c; c = t; }
    const int tri[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
    if (c < 3) return tri[r][c];
    if (r < 3) return c < 6 ? -1 : 6 + r * 3 + (c - 6);
    if (c < 6) return 15 + tri[r - 3][c - 3];
    if (r < 6) return 21 + (r - 3) * 3 + (c - 6);
    return 30 + tri[r - 6][c - 6];
}

template <class Gen>
__device__ void gram_accumulate(FinShared &sh, int N, Gen gen, double *dst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double acc[36];
#pragma unroll
    for (int q = 0; q < 36; ++q) acc[q] = 0.0;
    for (int n = threadIdx.x; n < N; n += kFinThreads) {
        double a[6], b[6];
        gen(n, a, b);
        acc[0] = fma(a[0], a[0], acc[0]); acc[1] = fma(a[0], a[1], acc[1]); acc[2] = fma(a[0], a[2], acc[2]);
        acc[3] = fma(a[1], a[1], acc[3]); acc[4] = fma(a[1], a[2], acc[4]); acc[5] = fma(a[2], a[2], acc[5]);
        acc[15] = fma(b[0], b[0], acc[15]); acc[16] = fma(b[0], b[1], acc[16]); acc[17] = fma(b[0], b[2], acc[17]);
        acc[18] = fma(b[1], b[1], acc[18]); acc[19] = fma(b[1], b[2], acc[19]); acc[20] = fma(b[2], b[2], acc[20]);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                acc[6 + r * 3 + c] = fma(a[r], a[3 + c], acc[6 + r * 3 + c]);
                acc[21 + r * 3 + c] = fma(b[r], b[3 + c], acc[21 + r * 3 + c]);
            }
        acc[30] = fma(a[3], a[3], fma(b[3], b[3], acc[30])); acc[31] = fma(a[3], a[4], fma(b[3], b[4], acc[31]));
        acc[32] = fma(a[3], a[5], fma(b[3], b[5], acc[32])); acc[33] = fma(a[4], a[4], fma(b[4], b[4], acc[33]));
        acc[34] = fma(a[4], a[5], fma(b[4], b[5], acc[34])); acc[35] = fma(a[5], a[5], fma(b[5], b[5], acc[35]));
    }
#pragma unroll
    for (int q = 0; q < 36; ++q) {
        const double sum = wave_sum(acc[q]);
        if (lane == 0) sh.gram[wave][q] = sum;
    }
    __syncthreads();
    if (threadIdx.x < 81) {
        const int slot = gram_slot(threadIdx.x / 9, threadIdx.x % 9);
        double t = 0;
        if (slot >= 0)
            for (int w = 0; w < kFinWaves; ++w) t += sh.gram[w][slot];
        dst[threadIdx.x] = t;
    }
    __syncthreads();
}

__device__ __forceinline__ double rl(double v, int srclane) { return lane_get<true>(v, srclane); }  // srclane is a compile-time constant at every call

// Null vector of the 9x9 DLT normal matrix on wave 0: lane r (< 9) holds row r of G.  The oracle (like OpenCV) runs a
// cyclic Jacobi eigen-decomposition and picks the eigenvector of the smallest eigenvalue; on the GPU that was 390 k cycles
// of dependent fp64 divisions, square roots and v_readlanes (a quarter of the finish kernel).  Only that one vector is
// needed, so: shift G by 1e-13 of its mean diagonal (keeps the factorisation away from an exactly singular matrix; the
// eigenvectors do not move), LU-factorise once (symmetric positive definite: no pivoting), and run inverse iteration,
// which converges to the same vector by a factor lambda_min/lambda_2 per step (<= 1e-3 on inlier sets).  The result
// agrees with the Jacobi vector up to the conditioning of G itself (H within 1e-5 px).  Returns h[lane].
__device__ double null_vector9(double (&M)[9], int lane) {
    double tr = 0;
#pragma unroll
    for (int j = 0; j < 9; ++j) tr += rl(M[j], j);
    const double mu = 1e-13 * tr / 9.0 + 1e-300;
#pragma unroll
    for (int j = 0; j < 9; ++j)
        if (lane == j) M[j] += mu;
    // in-place LU: after step c lane r > c keeps L[r][c] in M[c]; row r of U stays in M[r..8] of lane r
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const double inv = 1.0 / rl(M[c], c);
        const double f = M[c] * inv;
#pragma unroll
        for (int k = c + 1; k < 9; ++k) {
            const double prow = rl(M[k], c);
            if (lane > c) M[k] -= f * prow;
        }
        if (lane > c) M[c] = f;
    }
    double dinv = 1.0;  // 1 / U[lane][lane]
#pragma unroll
    for (int j = 0; j < 9; ++j)
        if (lane == j) dinv = 1.0 / M[j];
    double x = (lane < 9) ? 1.0 / 3.0 : 0.0, prev = x;
    for (int it = 0; it < 24; ++it) {
        double b = x;
#pragma unroll
        for (int c = 0; c < 8; ++c) {  // L y = x
            const double bc = rl(b, c);
            if (lane > c && lane < 9) b -= M[c] * bc;
        }
#pragma unroll
        for (int k = 8; k >= 0; --k) {  // U z = y
            const double xk = rl(b * dinv, k);
            if (lane == k) x = xk;
            if (lane < k) b -= M[k] * xk;
        }
        double n2 = 0, dot = 0;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const double xj = rl(x, j), pj = rl(prev, j);
            n2 += xj * xj;
            dot += xj * pj;
        }
        const double sc = (dot < 0 ? -1.0 : 1.0) / sqrt(n2);
        x = (lane < 9) ? x * sc : 0.0;
        double d2 = 0;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const double dj = rl(x, j) - rl(prev, j);
            d2 += dj * dj;
        }
        prev = x;
        if (d2 <= 1e-28) break;  // wave-uniform (every term came through v_readlane)
    }
    return x;
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
    const int *bound;       // adaptive RANSAC: (Bt) upper bounds from the head kernel, or null (all T hypotheses were scored)
    double confidence;
    int *iters_used;        // (Bt) or null
};

__global__ __launch_bounds__(kFinThreads) void finish_kernel(FinParams P) {
    __shared__ FinShared sh;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = P.N;
#ifdef GFN_ABLATE
    long long fin_t[5] = {(long long)__builtin_readcyclecounter(), 0, 0, 0, 0};
    int fin_lm = 0;
#define FIN_STAMP(i) fin_t[i] = (long long)__builtin_readcyclecounter()
#else
#define FIN_STAMP(i) do { } while (0)
#endif
    const float4 *pts = reinterpret_cast<const float4 *>(P.pts) + (size_t)b * N;
    const float *wgt = P.weight ? P.weight + (size_t)b * N : nullptr;
    unsigned char *mask = P.mask ? P.mask + (size_t)b * N : nullptr;

    int cnt = 0;
    if (P.mode == 0) {
        if (!P.bound) {
            // ---- all hypotheses scored: arg-max, key = (count << 32) | ~t  -> max count, lowest t (what the sequential loop
            // keeps: a later hypothesis must beat the best strictly) ----------
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
                sh.best = c > 3 ? (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFu)) : -1;  // goodCount > modelPoints - 1
                if (P.iters_used) P.iters_used[b] = N >= 4 ? P.T : 0;
            }
            __syncthreads();
        } else {
            // ---- adaptive RANSAC: replay cv::RANSACPointSetRegistrator::run over the counts in hypothesis order.  Counts
            // exist for t < max(kHeadHyp, bound[b]) (head kernel + gated hyp / score launches); the bound only ever shrinks, so
            // the loop stops inside that range (possibly inside the head, whose last update may even leave the bound below the
            // number of hypotheses already looked at).
            const int L = min(max(P.bound[b], kHeadHyp), P.T);
            int niters = N >= 4 ? P.T : 0, bestc = 0, best = -1;
            for (int c0 = 0; c0 < L; c0 += kFinScan) {  // block-uniform trip count
                for (int t = tid; t < kFinScan && c0 + t < L; t += kFinThreads) sh.scan[t] = P.counts[(size_t)b * P.T + c0 + t];
                __syncthreads();
                if (tid == 0) {
                    const int n = min(kFinScan, L - c0);
                    for (int k = 0; k < n; ++k) {
                        const int t = c0 + k;
                        if (t >= niters) break;
                        const int c = sh.scan[k];
                        if (c > (bestc > 3 ? bestc : 3)) {  // goodCount > max(maxGoodCount, modelPoints - 1)
                            bestc = c; best = t;
                            niters = ransac_update_iters(P.confidence, c, N, niters);
                        }
                    }
                }
                __syncthreads();
            }
            if (tid == 0) {
                sh.best = best;
                if (P.iters_used) P.iters_used[b] = niters;
            }
            __syncthreads();
        }
        const int best = sh.best;
        if (tid < 9) sh.H[tid] = best >= 0 ? P.hyp[((size_t)b * P.T + best) * 9 + tid] : (tid == 8 ? 1.0 : 0.0);
        __syncthreads();
        // ---- inlier mask of the best hypothesis ------------------------------------------------
        double Hb[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) Hb[k] = sh.H[k];
        int c = 0;
        for (int n = tid; n < N; n += kFinThreads) {
            const bool in = best >= 0 && is_inlier(Hb, pts[n], P.thr2);
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
        if (P.mode == 0) return is_inlier(Hb, pts[n], P.thr2) ? 1.0 : 0.0;
        return wgt ? (double)wgt[n] : 1.0;
    };
    double Hb[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) Hb[k] = (P.mode == 0) ? sh.H[k] : 0.0;
    FIN_STAMP(1);

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
        gram_accumulate(sh, N, [&](int n, double (&a)[6], double (&bb)[6]) {
            const double sw_ = sqrt(point_w(n, Hb));
            const float4 p = pts[n];
            const double X = (p.x - cx) * sx * sw_, Y = (p.y - cy) * sy * sw_, x = (p.z - cu) * su, y = (p.w - cv) * sv;
            a[0] = X; a[1] = Y; a[2] = sw_; a[3] = -x * X; a[4] = -x * Y; a[5] = -x * sw_;
            bb[0] = X; bb[1] = Y; bb[2] = sw_; bb[3] = -y * X; bb[4] = -y * Y; bb[5] = -y * sw_;
        }, sh.G);
        // eigenvector of the smallest eigenvalue (wave 0, one matrix row per lane)
        if (wave == 0) {
            double A[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) A[j] = (lane < 9) ? sh.G[lane * 9 + j] : 0.0;
            const double hk = null_vector9(A, lane);
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
        gram_accumulate(sh, N, [&](int n, double (&a)[6], double (&bb)[6]) {
            const double w = point_w(n, Hb);  // 0/1 here
            const float4 p = pts[n];
            const double X = p.x, Y = p.y, u = p.z, v = p.w;
            const double ww = w != 0.0 ? 1.0 / (h[6] * X + h[7] * Y + 1.0) : 0.0;  // masked points contribute exact zeros
            const double xi = (h[0] * X + h[1] * Y + h[2]) * ww, yi = (h[3] * X + h[4] * Y + h[5]) * ww;
            // xi, yi carry the weight; for w = 1 these are the reference's Jacobian rows and residuals
            a[0] = X * ww; a[1] = Y * ww; a[2] = ww; a[3] = -X * ww * xi; a[4] = -Y * ww * xi; a[5] = xi - w * u;
            bb[0] = X * ww; bb[1] = Y * ww; bb[2] = ww; bb[3] = -X * ww * yi; bb[4] = -Y * ww * yi; bb[5] = yi - w * v;
        }, dst);
    };
    if (tid < 9) sh.h[tid] = sh.H[tid] / sh.H[8];
    __syncthreads();
    FIN_STAMP(2);
    lm_gram(sh.h, sh.G);
    double S = sh.G[80], lambda = 1e-3;
    for (int it = 0; it < P.lm_iters; ++it) {
#ifdef GFN_ABLATE
        ++fin_lm;
#endif
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
            // largest step component, for cv::LMSolver's stopping rule (epsx = FLT_EPSILON)
            double dmax = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) dmax = fmax(dmax, fabs(lane_get<true>(d, k)));
            if (lane == 0) { sh.flag = ok ? 1 : 0; sh.stats[10] = dmax; }
        }
        __syncthreads();
        if (!sh.flag) { lambda *= 10; __syncthreads(); continue; }
        lm_gram(sh.hn, sh.G2);
        const double S2 = sh.G2[80];
        const bool accept = S2 < S;
        const bool stop = sh.stats[10] < 1.1920928955078125e-07;  // cv::LMSolver: the step just tried (accepted or not) is below epsx
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
    FIN_STAMP(3);
#ifdef GFN_ABLATE
    if (tid == 0 && b < 3)
        printf("finish b%d: select+mask %lld | DLT %lld | LM %lld cycles, %d LM iterations, %d inliers\n", b, fin_t[1] - fin_t[0], fin_t[2] - fin_t[1],
               fin_t[3] - fin_t[2], fin_lm, cnt);
#endif
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
    // hypotheses (9 doubles) + counts (int) + the per-pair iteration bounds of the adaptive rule, rounded up
    return (int64_t)Bt * iters * (9 * 8 + 4) + (int64_t)Bt * 4 + 256;
}

GFN_EXPORT int gfn_homography_ransac(const float *pts, int Bt, int N, double thresh, int iters, uint64_t seed, int lm_iters,
                                     int stage, double *H, int *ninl, int *best_t, unsigned char *mask, void *scratch,
                                     int64_t scratch_bytes, gfn_stream_t stream) {
    return gfn_homography_ransac_ex(pts, Bt, N, thresh, iters, 0.0, seed, lm_iters, stage, H, ninl, best_t, mask, nullptr, scratch, scratch_bytes,
                                    stream);
}

GFN_EXPORT int gfn_homography_ransac_ex(const float *pts, int Bt, int N, double thresh, int iters, double confidence, uint64_t seed,
                                        int lm_iters, int stage, double *H, int *ninl, int *best_t, unsigned char *mask, int *iters_used,
                                        void *scratch, int64_t scratch_bytes, gfn_stream_t stream) {
    if (!pts || !H || !ninl || !best_t) return gfn::fail(GFN_ERR_INVALID_ARG, "homography_ransac: null pointer");
    if (Bt < 0 || N < 0 || iters <= 0 || !(thresh > 0) || lm_iters < 0 || stage < 0 || stage > 2 || !(confidence < 1.0))
        return gfn::fail(GFN_ERR_INVALID_ARG, "homography_ransac: bad argument");
    if ((uintptr_t)pts & 15) return gfn::fail(GFN_ERR_INVALID_ARG, "homography_ransac: pts must be 16-byte aligned");
    if (!scratch || scratch_bytes < gfn_homography_scratch_bytes(Bt, iters))
        return gfn::fail(GFN_ERR_SCRATCH, "homography_ransac: scratch too small (%lld < %lld bytes)", (long long)scratch_bytes,
                         (long long)gfn_homography_scratch_bytes(Bt, iters));
    if (Bt == 0) return GFN_OK;
    hipStream_t s = (hipStream_t)stream;
    double *hyp = reinterpret_cast<double *>(scratch);
    int *counts = reinterpret_cast<int *>(hyp + (size_t)Bt * iters * 9);
    int *bound = nullptr;
    int tbeg = 0;
    if (confidence > 0) {
        // the iteration bound follows the best inlier ratio (OpenCV's control flow): the head of the loop per pair, then only
        // the hypotheses the sequential loop can still reach
        bound = counts + (size_t)Bt * iters;
        hipLaunchKernelGGL(ransac_head_kernel, dim3(Bt), dim3(kHeadThreads), 0, s, pts, hyp, counts, bound, N, iters, seed, thresh * thresh,
                           confidence);
        if (int e = gfn::check_launch("ransac_head_kernel")) return e;
        tbeg = kHeadHyp;
    }
    if (iters > tbeg) {
        const long nh = (long)Bt * (iters - tbeg);
        hipLaunchKernelGGL(hyp_kernel, dim3((unsigned)((nh * 8 + 255) / 256)), dim3(256), 0, s, pts, hyp, Bt, N, iters, tbeg, bound, seed);
        if (int e = gfn::check_launch("hyp_kernel")) return e;
        const long ngroups = (long)Bt * ((iters - tbeg + kHypPerWave - 1) / kHypPerWave);
        hipLaunchKernelGGL(score_kernel, dim3((unsigned)((ngroups + 3) / 4)), dim3(256), 0, s, pts, hyp, counts, Bt, N, iters, tbeg, bound,
                           thresh * thresh);
        if (int e = gfn::check_launch("score_kernel")) return e;
    }
    FinParams P;
    P.pts = pts; P.weight = nullptr; P.hyp = hyp; P.counts = counts; P.H = H; P.ninl = ninl; P.best_t = best_t; P.mask = mask;
    P.N = N; P.T = iters; P.thr2 = thresh * thresh; P.lm_iters = lm_iters; P.stage = stage; P.mode = 0;
    P.bound = bound; P.confidence = confidence; P.iters_used = iters_used;
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
    P.bound = nullptr; P.confidence = 0; P.iters_used = nullptr;
    hipLaunchKernelGGL(finish_kernel, dim3(Bt), dim3(kFinThreads), 0, (hipStream_t)stream, P);
    return gfn::check_launch("finish_kernel(dlt)");
}
