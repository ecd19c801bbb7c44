/*
 * homography_oracle.c -- CPU restatement of the homography solve.  TEST INFRASTRUCTURE ONLY.
 *
 * Reference call site: estimation.py:60-77 -- cv2.findHomography(pos_a, pos_b, cv2.RANSAC,
 * confidence=0.99999, ransacReprojThreshold=3).  OpenCV is a third-party dependency that is NOT
 * part of the reference repository and is not installed here (opencv-python, version unpinned in
 * requirements.txt:2), so this file restates the *published* findHomography pipeline
 *   RANSAC over 4-point minimal sets (subsets with collinear points or inconsistent orientation are re-drawn,
 *   HomographyEstimatorCallback::checkSubset), reprojection threshold on squared error, the iteration bound shrinking
 *   with the best inlier ratio (RANSACUpdateNumIters: niters = log(1 - confidence) / log(1 - w^4))
 *   -> normalised DLT on the inliers (centroid / mean-absolute-deviation normalisation, 9x9
 *      L^T L, eigenvector of the smallest eigenvalue, de-normalise, H /= H[2][2])
 *   -> Levenberg-Marquardt refinement of the 8 free parameters on the inliers (<= 10 iterations)
 * with a counter-based RNG of our own (OpenCV's RNG stream cannot be reproduced).
 * PARITY WITH OPENCV IS THEREFORE UNPINNED; this oracle is pinned by known-H synthetic
 * correspondences (tests/test_homography_cpu.py) and is the checker for the HIP solver, which
 * follows exactly the same algorithm (same RNG, same hypothesis order, same tie-breaking).
 *
 * All arithmetic is double; built with -ffp-contract=off so that the per-point reprojection test
 * rounds identically on the GPU (hipcc -ffp-contract=off): hypotheses, inlier counts and the
 * chosen hypothesis are bit-identical, the later reductions agree to ~1e-15 relative.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define EXPORT __attribute__((visibility("default")))

static inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* index k (0..3) of hypothesis t of pair b, attempt a at the slot, subset number s of the hypothesis */
static inline uint32_t draw_index(uint64_t seed, uint32_t b, uint32_t t, uint32_t k, uint32_t a, uint32_t s, uint32_t N) {
    uint64_t h = splitmix64(seed ^ splitmix64(((uint64_t)b << 32) | t));
    h = splitmix64(h + ((uint64_t)s << 16) + ((uint64_t)k << 8) + a);
    return (uint32_t)(h % N);
}

/* four distinct indices; returns 0 if that failed after 16 attempts per slot */
static int draw_sample(uint64_t seed, uint32_t b, uint32_t t, uint32_t s, uint32_t N, uint32_t idx[4]) {
    for (uint32_t k = 0; k < 4; ++k) {
        uint32_t a = 0;
        for (;;) {
            uint32_t v = draw_index(seed, b, t, k, a, s, N);
            int dup = 0;
            for (uint32_t q = 0; q < k; ++q) dup |= (idx[q] == v);
            if (!dup) { idx[k] = v; break; }
            if (++a >= 16) return 0;
        }
    }
    return 1;
}

/* HomographyEstimatorCallback::checkSubset for 4 correspondences: (a) haveCollinearPoints on either image -- the last point
 * must not lie on (or within rounding of) a line through two of the first three, nor coincide with them; (b) the four
 * triplets must keep or all flip their orientation between the images ("Speeding-up homography estimation in mobile
 * devices").  Plain fp64 operations in a fixed order: the GPU evaluates the same sequence. */
#define ORACLE_FLT_EPSILON 1.1920928955078125e-07
static int subset_ok(const float *pts, const uint32_t idx[4]) {
    double X[2][4], Y[2][4];
    for (int k = 0; k < 4; ++k) {
        const float *p = pts + 4 * (size_t)idx[k];
        X[0][k] = p[0]; Y[0][k] = p[1]; X[1][k] = p[2]; Y[1][k] = p[3];
    }
    for (int im = 0; im < 2; ++im)
        for (int j = 0; j < 3; ++j) {
            const double dx1 = X[im][j] - X[im][3], dy1 = Y[im][j] - Y[im][3];
            for (int k = 0; k < j; ++k) {
                const double dx2 = X[im][k] - X[im][3], dy2 = Y[im][k] - Y[im][3];
                if (fabs(dx2 * dy1 - dy2 * dx1) <= ORACLE_FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2))) return 0;
            }
        }
    static const int tt[4][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
    int negative = 0;
    for (int i = 0; i < 4; ++i) {
        double det[2];
        for (int im = 0; im < 2; ++im) {
            const double x0 = X[im][tt[i][0]], y0 = Y[im][tt[i][0]], x1 = X[im][tt[i][1]], y1 = Y[im][tt[i][1]];
            const double x2 = X[im][tt[i][2]], y2 = Y[im][tt[i][2]];
            det[im] = (x0 * (y1 - y2) - y0 * (x1 - x2)) + (x1 * y2 - x2 * y1);
        }
        negative += (det[0] * det[1] < 0);
    }
    return negative == 0 || negative == 4;
}

/* the subset of hypothesis t: up to ORACLE_SUBSET_TRIES draws until one passes checkSubset (OpenCV re-draws inside
 * getSubset); subset 0 is the stream of the earlier fixed-iteration solver */
#define ORACLE_SUBSET_TRIES 8
static int draw_checked(const float *pts, uint64_t seed, uint32_t b, uint32_t t, uint32_t N, uint32_t idx[4]) {
    for (uint32_t s = 0; s < ORACLE_SUBSET_TRIES; ++s) {
        if (!draw_sample(seed, b, t, s, N, idx)) continue;
        if (subset_ok(pts, idx)) return 1;
    }
    return 0;
}

/* log(x), x > 0 and normal, from + - * / only (no libm): the CPU and the GPU must agree on every bit of the iteration bound.
 * x = m 2^e with m in [1/sqrt2, sqrt2); log m = 2 atanh((m-1)/(m+1)), 15 odd terms (|t| <= 0.172: t^31 < 1e-23). */
static double det_log(double x) {
    uint64_t bits;
    memcpy(&bits, &x, 8);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    bits = (bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m;
    memcpy(&m, &bits, 8);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double t = (m - 1.0) / (m + 1.0), t2 = t * t;
    double s = 0.0;
    for (int k = 14; k >= 0; --k) s = s * t2 + 1.0 / (double)(2 * k + 1);
    return 2.0 * t * s + (double)e * 0.6931471805599453;
}

/* cv::RANSACUpdateNumIters(confidence, outlier ratio, modelPoints = 4, current bound) */
static int ransac_update_iters(double conf, int good, int N, int niters) {
    double p = conf < 0 ? 0 : (conf > 1 ? 1 : conf);
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

/* Solve the n x n system M x = rhs in place (Gaussian elimination, partial pivoting).
 * M is row-major with leading dimension n+1 (augmented).  Returns 0 if singular. */
static int solve_aug(double *M, int n) {
    const int ld = n + 1;
    for (int c = 0; c < n; ++c) {
        int piv = c;
        double best = fabs(M[c * ld + c]);
        for (int r = c + 1; r < n; ++r) {
            double v = fabs(M[r * ld + c]);
            if (v > best) { best = v; piv = r; }
        }
        if (!(best > 1e-300)) return 0;
        if (piv != c)
            for (int k = 0; k < ld; ++k) { double tmp = M[c * ld + k]; M[c * ld + k] = M[piv * ld + k]; M[piv * ld + k] = tmp; }
        const double inv = 1.0 / M[c * ld + c];
        for (int r = c + 1; r < n; ++r) {
            const double f = M[r * ld + c] * inv;
            for (int k = c; k < ld; ++k) M[r * ld + k] = M[r * ld + k] - f * M[c * ld + k];
        }
    }
    for (int r = n - 1; r >= 0; --r) {
        double s = M[r * ld + n];
        for (int k = n - 1; k > r; --k) s = s - M[r * ld + k] * M[k * ld + n]; /* descending: the GPU's lane-per-row order */
        M[r * ld + n] = s / M[r * ld + r];
    }
    return 1;
}

/* exact homography through 4 correspondences (h33 = 1); pts rows are (x,y,u,v) */
static int solve4(const float *pts, const uint32_t idx[4], double H[9]) {
    double M[8 * 9];
    for (int k = 0; k < 4; ++k) {
        const float *p = pts + 4 * (size_t)idx[k];
        const double x = p[0], y = p[1], u = p[2], v = p[3];
        double *r0 = M + (2 * k) * 9, *r1 = M + (2 * k + 1) * 9;
        r0[0] = x; r0[1] = y; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -u * x; r0[7] = -u * y; r0[8] = u;
        r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = x; r1[4] = y; r1[5] = 1; r1[6] = -v * x; r1[7] = -v * y; r1[8] = v;
    }
    if (!solve_aug(M, 8)) return 0;
    for (int k = 0; k < 8; ++k) H[k] = M[k * 9 + 8];
    H[8] = 1.0;
    for (int k = 0; k < 8; ++k)
        if (!isfinite(H[k])) return 0;
    return 1;
}

/* Inlier test of one correspondence: squared reprojection error <= thr2 (OpenCV HomographyEstimatorCallback::
 * computeError followed by the threshold test of RANSAC).  With (X, Y, w) = H (x, y, 1)^T the error is
 * (X/w - u)^2 + (Y/w - v)^2; multiplied through by w^2 the test reads (X - u w)^2 + (Y - v w)^2 <= thr2 w^2 -- no
 * division (w = 0 never passes).  The sums are written as explicit fused multiply-adds (fma() is exactly rounded whether
 * libm or the hardware evaluates it): the GPU kernels evaluate exactly this sequence of operations in fp64 -- 12 instead of
 * 25 instructions per test there -- so inlier counts and masks are bit-identical. */
static inline int is_inlier(const double H[9], const float *p, double thr2) {
    const double x = p[0], y = p[1], u = p[2], v = p[3];
    const double w = fma(H[6], x, fma(H[7], y, H[8]));
    const double dx = fma(-u, w, fma(H[0], x, fma(H[1], y, H[2])));
    const double dy = fma(-v, w, fma(H[3], x, fma(H[4], y, H[5])));
    const double w2 = w * w;
    return fma(dx, dx, dy * dy) <= thr2 * w2 && w2 > 0;
}

/* cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (n <= 9): A -> diag, V columns = eigenvectors */
static void jacobi_eig(double *A, double *V, int n) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[i * n + j] = (i == j);
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0, diag = 0;
        for (int i = 0; i < n; ++i) {
            diag += A[i * n + i] * A[i * n + i];
            for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
        }
        if (off <= 1e-30 * diag || off == 0) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[p * n + q];
                if (apq == 0) continue;
                const double theta = (A[q * n + q] - A[p * n + p]) / (2 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                const double c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
}

/* Weighted normalised DLT (OpenCV runKernel with per-point weights; weight NULL = 1, and only
 * points with mask != 0 when a mask is given).  Returns 0 on failure. */
static int dlt_normalised(const float *pts, const double *weight, const unsigned char *mask, int N, double H[9]) {
    double sw = 0, cx = 0, cy = 0, cu = 0, cv = 0;
    for (int n = 0; n < N; ++n) {
        if (mask && !mask[n]) continue;
        const double w = weight ? weight[n] : 1.0;
        sw += w; cx += w * pts[4 * n]; cy += w * pts[4 * n + 1]; cu += w * pts[4 * n + 2]; cv += w * pts[4 * n + 3];
    }
    if (!(sw > 0)) return 0;
    cx /= sw; cy /= sw; cu /= sw; cv /= sw;
    double ax = 0, ay = 0, au = 0, av = 0;
    for (int n = 0; n < N; ++n) {
        if (mask && !mask[n]) continue;
        const double w = weight ? weight[n] : 1.0;
        ax += w * fabs(pts[4 * n] - cx); ay += w * fabs(pts[4 * n + 1] - cy);
        au += w * fabs(pts[4 * n + 2] - cu); av += w * fabs(pts[4 * n + 3] - cv);
    }
    if (!(ax > 1e-300 && ay > 1e-300 && au > 1e-300 && av > 1e-300)) return 0;
    const double sx = sw / ax, sy = sw / ay, su = sw / au, sv = sw / av;
    double L[81];
    memset(L, 0, sizeof(L));
    for (int n = 0; n < N; ++n) {
        if (mask && !mask[n]) continue;
        const double w = weight ? weight[n] : 1.0;
        const double X = (pts[4 * n] - cx) * sx, Y = (pts[4 * n + 1] - cy) * sy;
        const double x = (pts[4 * n + 2] - cu) * su, y = (pts[4 * n + 3] - cv) * sv;
        const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
        const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
        for (int i = 0; i < 9; ++i)
            for (int j = i; j < 9; ++j) L[i * 9 + j] += w * (Lx[i] * Lx[j] + Ly[i] * Ly[j]);
    }
    for (int i = 0; i < 9; ++i)
        for (int j = 0; j < i; ++j) L[i * 9 + j] = L[j * 9 + i];
    double V[81];
    jacobi_eig(L, V, 9);
    int kmin = 0;
    for (int k = 1; k < 9; ++k)
        if (L[k * 9 + k] < L[kmin * 9 + kmin]) kmin = k;
    double h[9];
    for (int k = 0; k < 9; ++k) h[k] = V[k * 9 + kmin];
    /* H = inv(T_dst) * H0 * T_src,  T_src = [sx 0 -cx*sx; 0 sy -cy*sy; 0 0 1], inv(T_dst) = [1/su 0 cu; 0 1/sv cv; 0 0 1] */
    double A[9];
    for (int c = 0; c < 3; ++c) {
        A[0 + c] = h[0 + c] / su + cu * h[6 + c];
        A[3 + c] = h[3 + c] / sv + cv * h[6 + c];
        A[6 + c] = h[6 + c];
    }
    for (int r = 0; r < 3; ++r) {
        H[3 * r + 0] = A[3 * r + 0] * sx;
        H[3 * r + 1] = A[3 * r + 1] * sy;
        H[3 * r + 2] = A[3 * r + 2] - A[3 * r + 0] * cx * sx - A[3 * r + 1] * cy * sy;
    }
    if (!(fabs(H[8]) > 1e-300)) return 0;
    const double inv = 1.0 / H[8];
    for (int k = 0; k < 9; ++k) H[k] *= inv;
    for (int k = 0; k < 9; ++k)
        if (!isfinite(H[k])) return 0;
    return 1;
}

/* normal equations of the reprojection residuals at h (h[8] == 1): G = [J|r]^T [J|r] (9x9) */
static void lm_gram(const float *pts, const unsigned char *mask, int N, const double h[9], double G[81]) {
    memset(G, 0, 81 * sizeof(double));
    for (int n = 0; n < N; ++n) {
        if (mask && !mask[n]) continue;
        const double X = pts[4 * n], Y = pts[4 * n + 1], u = pts[4 * n + 2], v = pts[4 * n + 3];
        const double ww = 1.0 / (h[6] * X + h[7] * Y + 1.0);
        const double xi = (h[0] * X + h[1] * Y + h[2]) * ww, yi = (h[3] * X + h[4] * Y + h[5]) * ww;
        const double Jx[9] = {X * ww, Y * ww, ww, 0, 0, 0, -X * ww * xi, -Y * ww * xi, xi - u};
        const double Jy[9] = {0, 0, 0, X * ww, Y * ww, ww, -X * ww * yi, -Y * ww * yi, yi - v};
        for (int i = 0; i < 9; ++i)
            for (int j = i; j < 9; ++j) G[i * 9 + j] += Jx[i] * Jx[j] + Jy[i] * Jy[j];
    }
    for (int i = 0; i < 9; ++i)
        for (int j = 0; j < i; ++j) G[i * 9 + j] = G[j * 9 + i];
}

static void lm_refine(const float *pts, const unsigned char *mask, int N, double H[9], int iters) {
    double h[9], G[81];
    for (int k = 0; k < 9; ++k) h[k] = H[k] / H[8];
    lm_gram(pts, mask, N, h, G);
    double S = G[80], lambda = 1e-3;
    for (int it = 0; it < iters; ++it) {
        double M[8 * 9];
        for (int i = 0; i < 8; ++i) {
            for (int j = 0; j < 8; ++j) M[i * 9 + j] = G[i * 9 + j];
            M[i * 9 + i] = G[i * 9 + i] + lambda * G[i * 9 + i];
            M[i * 9 + 8] = -G[i * 9 + 8];
        }
        if (!solve_aug(M, 8)) { lambda *= 10; continue; }
        double hn[9], G2[81], dmax = 0;
        for (int k = 0; k < 8; ++k) { hn[k] = h[k] + M[k * 9 + 8]; if (fabs(M[k * 9 + 8]) > dmax) dmax = fabs(M[k * 9 + 8]); }
        hn[8] = 1.0;
        lm_gram(pts, mask, N, hn, G2);
        if (G2[80] < S) {
            memcpy(h, hn, sizeof(h));
            memcpy(G, G2, sizeof(G));
            S = G2[80];
            lambda = lambda > 1e-11 ? lambda / 10 : 1e-12;
        } else {
            lambda *= 10;
        }
        /* cv::LMSolverImpl::run (createLMSolver(cb, 10), epsx = FLT_EPSILON): proceed while the largest component of the step
         * just tried -- accepted or not -- is at least epsx; on a matcher's inliers that is three or four iterations, not ten */
        if (dmax < ORACLE_FLT_EPSILON) break;
    }
    memcpy(H, h, sizeof(h));
}

/* estimation.py:26-45 convert_coordinates in float32 exactly as numpy evaluates it:
 * (w-1) * (x+1) / 2 on a float32 array with python-int scalars. */
EXPORT void oracle_convert_matches(const float *matches, float *pts, long n, float wA, float hA, float wB, float hB) {
    for (long i = 0; i < n; ++i) {
        pts[4 * i + 0] = ((wA - 1.f) * (matches[4 * i + 0] + 1.f)) / 2.f;
        pts[4 * i + 1] = ((hA - 1.f) * (matches[4 * i + 1] + 1.f)) / 2.f;
        pts[4 * i + 2] = ((wB - 1.f) * (matches[4 * i + 2] + 1.f)) / 2.f;
        pts[4 * i + 3] = ((hB - 1.f) * (matches[4 * i + 3] + 1.f)) / 2.f;
    }
}

/* One-shot weighted DLT ("grid-DLT"): pts (Bt,N,4) pixel coordinates, weight (Bt,N) or NULL.
 * H (Bt,9) row-major, ok (Bt). */
EXPORT void oracle_homography_dlt(const float *pts, const double *weight, int Bt, int N, double *H, int *ok) {
#pragma omp parallel for schedule(dynamic)
    for (int b = 0; b < Bt; ++b) {
        double h[9];
        int good = dlt_normalised(pts + (size_t)b * N * 4, weight ? weight + (size_t)b * N : NULL, NULL, N, h);
        if (!good) { memset(h, 0, sizeof(h)); h[8] = 1.0; }
        memcpy(H + 9 * (size_t)b, h, sizeof(h));
        ok[b] = good;
    }
}

/* RANSAC -> DLT on inliers -> LM.  pts (Bt,N,4); H (Bt,9); ninl (Bt); best_t (Bt) the chosen
 * hypothesis index (or -1); mask (Bt,N) or NULL; stage: 0 = full pipeline, 1 = stop after RANSAC
 * (H = best minimal-sample hypothesis), 2 = stop after the inlier DLT.
 * confidence in (0,1): cv::RANSACPointSetRegistrator::run -- whenever a hypothesis beats the best inlier count so far the
 * iteration bound becomes RANSACUpdateNumIters(confidence, ...) (estimation.py:66-72 passes 0.99999); confidence <= 0: all
 * `iters` hypotheses are scored.  iters_used (Bt) or NULL: the bound the loop stopped at. */
EXPORT void oracle_homography_ransac(const float *pts, int Bt, int N, double thresh, int iters, double confidence, uint64_t seed,
                                     int lm_iters, int stage, double *H, int *ninl, int *best_t, unsigned char *mask, int *iters_used) {
    const double thr2 = thresh * thresh;
#pragma omp parallel for schedule(dynamic)
    for (int b = 0; b < Bt; ++b) {
        const float *p = pts + (size_t)b * N * 4;
        int best = -1, bestc = 0, niters = iters;
        double Hb[9] = {0, 0, 0, 0, 0, 0, 0, 0, 1};
        for (int t = 0; t < niters && N >= 4; ++t) {
            uint32_t idx[4];
            double Ht[9];
            if (!draw_checked(p, seed, (uint32_t)b, (uint32_t)t, (uint32_t)N, idx)) continue;
            if (!solve4(p, idx, Ht)) continue;
            int c = 0;
            for (int n = 0; n < N; ++n) c += is_inlier(Ht, p + 4 * n, thr2);
            if (c > (bestc > 3 ? bestc : 3)) {  /* goodCount > max(maxGoodCount, modelPoints - 1) */
                bestc = c; best = t; memcpy(Hb, Ht, sizeof(Hb));
                if (confidence > 0) niters = ransac_update_iters(confidence, c, N, niters);
            }
        }
        if (iters_used) iters_used[b] = N >= 4 ? niters : 0;
        unsigned char *mk = (unsigned char *)malloc(N > 0 ? N : 1);
        int cnt = 0;
        for (int n = 0; n < N; ++n) { mk[n] = best >= 0 && is_inlier(Hb, p + 4 * n, thr2); cnt += mk[n]; }
        if (best < 0 || cnt < 4) {
            memset(Hb, 0, sizeof(Hb));
            Hb[8] = 1.0; /* estimation.py:74-76: failure -> diag(0,0,1) */
            best = -1;
        } else if (stage != 1) {
            double Hd[9];
            if (cnt > 4 && dlt_normalised(p, NULL, mk, N, Hd)) memcpy(Hb, Hd, sizeof(Hb));
            if (stage != 2 && cnt > 4) lm_refine(p, mk, N, Hb, lm_iters);
        }
        memcpy(H + 9 * (size_t)b, Hb, sizeof(Hb));
        ninl[b] = cnt;
        best_t[b] = best;
        if (mask) memcpy(mask + (size_t)b * N, mk, N);
        free(mk);
    }
}
