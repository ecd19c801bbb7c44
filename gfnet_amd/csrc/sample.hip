// sample.hip -- weighted sampling without replacement for GFNet.sample on gfx950.
//
// Replaces the two torch.multinomial(p, num_samples, replacement=False) draws of GFNet.sample
// (model/network.py:400-402 and :411-413).  torch implements that draw as an "exponential race": every
// element gets the key w_i / E_i with E_i ~ Exp(1) and the num_samples largest keys win; the library then
// spends 0.55 ms per 32-pair batch in a multi-block top-k plus a merge sort of the winners (the order of the
// winners is irrelevant to the caller).  Here:
//   race_keys_kernel     one thread per element: counter-based uniform (splitmix64 of seed, row, index),
//                        key = w / -log(u) as order-preserving 32-bit pattern (w <= 0 -> key 0)
//   race_select_kernel   one 1024-thread workgroup per row: three radix-select passes (11 + 11 + 10 bits,
//                        LDS histograms) find the exact k-th largest key, then one ordered pass writes the
//                        indices of the winners in increasing index order (ties at the threshold are taken
//                        in index order) -- deterministic for a given seed, no sort, no atomics on the output.
// Same distribution as torch.multinomial without replacement; not the same random stream.
#include "common.h"

namespace {

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// One workgroup = kKeysPerBlock consecutive elements of one row; besides the keys it accumulates the row's histogram of
// the top 11 key bits (the first radix-select pass) -- in LDS, flushed with one global atomic per occupied bin.
constexpr int kKeysPerBlock = 8192;
__global__ __launch_bounds__(256) void race_keys_kernel(const float *__restrict__ w, long w_rs, unsigned *__restrict__ keys,
                                                        unsigned *__restrict__ hist1, int N, int blocks_per_row, uint64_t seed,
                                                        float one_above) {
    __shared__ unsigned hist[2048];
    const int row = blockIdx.x / blocks_per_row, chunk = blockIdx.x - row * blocks_per_row;
    for (int e = threadIdx.x; e < 2048; e += 256) hist[e] = 0;
    __syncthreads();
    const int i0 = chunk * kKeysPerBlock, i1 = min(N, i0 + kKeysPerBlock);
    for (int i = i0 + threadIdx.x; i < i1; i += 256) {
        float wi = w[(size_t)row * w_rs + i];
        wi = wi > one_above ? 1.f : wi;  // GFNet.sample's certainty threshold (network.py:391-393); +inf = off
        const uint64_t h = splitmix64(seed ^ splitmix64(((uint64_t)(unsigned)row << 32) | (unsigned)i));
        const float u = ((float)(unsigned)(h >> 40) + 1.0f) * 5.9604644775390625e-08f;  // (0, 1], 24 bits
        const float e = -__logf(u);                                                      // Exp(1); 0 only for u == 1
        float key = wi > 0.f ? wi / fmaxf(e, 1e-30f) : 0.f;
        key = key < 3.0e38f ? key : 3.0e38f;
        const unsigned k = __float_as_uint(key);  // non-negative floats order like their bit patterns
        keys[(size_t)row * N + i] = k;
        atomicAdd(&hist[k >> 21], 1u);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2048; e += 256)
        if (hist[e]) atomicAdd(&hist1[(size_t)row * 2048 + e], hist[e]);
}

constexpr int kSelThreads = 1024;
constexpr int kSelWaves = kSelThreads / 64;

// Find, among the elements whose key matches `prefix` in the bits above `shift + bits`, the digit (bits wide, at `shift`)
// that contains the need-th largest; returns it and lowers `need` by the elements in larger digits.  All threads call it.
template <int BITS>
__device__ unsigned radix_pass(const unsigned *__restrict__ keys, int N, unsigned prefix_mask, unsigned prefix, int shift, unsigned &need,
                               unsigned *hist, unsigned *bcast, const unsigned *__restrict__ prefilled = nullptr) {
    constexpr int BINS = 1 << BITS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < BINS; e += kSelThreads) hist[e] = prefilled ? prefilled[e] : 0;
    __syncthreads();
    if (!prefilled) {
        // a pass is a chain of L2 round trips for the single workgroup of a row: 16-byte loads, 4 in flight per thread
        // (4 keys each) when the row allows, else 8 scalar loads
        const bool vec = ((N & 3) == 0) && ((reinterpret_cast<uintptr_t>(keys) & 15) == 0);
        if (vec) {
            constexpr int UF = 4;
            const uint4 *k4 = reinterpret_cast<const uint4 *>(keys);
            const int N4 = N >> 2;
            for (int n0 = tid; n0 < N4; n0 += UF * kSelThreads) {
                uint4 k[UF];
#pragma unroll
                for (int q = 0; q < UF; ++q) k[q] = n0 + q * kSelThreads < N4 ? k4[n0 + q * kSelThreads] : make_uint4(~prefix, ~prefix, ~prefix, ~prefix);
#pragma unroll
                for (int q = 0; q < UF; ++q) {
                    if (n0 + q * kSelThreads < N4) {
                        if ((k[q].x & prefix_mask) == prefix) atomicAdd(&hist[(k[q].x >> shift) & (BINS - 1)], 1u);
                        if ((k[q].y & prefix_mask) == prefix) atomicAdd(&hist[(k[q].y >> shift) & (BINS - 1)], 1u);
                        if ((k[q].z & prefix_mask) == prefix) atomicAdd(&hist[(k[q].z >> shift) & (BINS - 1)], 1u);
                        if ((k[q].w & prefix_mask) == prefix) atomicAdd(&hist[(k[q].w >> shift) & (BINS - 1)], 1u);
                    }
                }
            }
        } else {
            constexpr int UF = 8;
            for (int n0 = tid; n0 < N; n0 += UF * kSelThreads) {
                unsigned k[UF];
#pragma unroll
                for (int q = 0; q < UF; ++q) k[q] = n0 + q * kSelThreads < N ? keys[n0 + q * kSelThreads] : ~prefix;
#pragma unroll
                for (int q = 0; q < UF; ++q)
                    if (n0 + q * kSelThreads < N && (k[q] & prefix_mask) == prefix) atomicAdd(&hist[(k[q] >> shift) & (BINS - 1)], 1u);
            }
        }
    }
    __syncthreads();
    if (wave == 0) {  // suffix sums from the top: lane l owns bins [l*PER, (l+1)*PER)
        constexpr int PER = BINS / 64;
        unsigned own = 0;
        for (int q = 0; q < PER; ++q) own += hist[lane * PER + q];
        unsigned above = 0;  // elements in lanes > l
        {
            unsigned incl = own;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned v = __shfl_down(incl, o);
                if (lane + o < 64) incl += v;
            }
            above = incl - own;
        }
        // the lane whose range crosses `need`
        const bool mine = above < need && above + own >= need;
        if (mine) {
            unsigned run = above;
            for (int q = PER - 1; q >= 0; --q) {
                const unsigned c = hist[lane * PER + q];
                if (run + c >= need) {
                    bcast[0] = (unsigned)(lane * PER + q);
                    bcast[1] = need - run;  // still needed inside this digit
                    break;
                }
                run += c;
            }
        }
    }
    __syncthreads();
    const unsigned digit = bcast[0];
    need = bcast[1];
    __syncthreads();
    return digit;
}

__global__ __launch_bounds__(kSelThreads) void race_select_kernel(const unsigned *__restrict__ keys_all, const unsigned *__restrict__ hist1,
                                                                  long long *__restrict__ out, int N, int K) {
    __shared__ unsigned hist[2048];
    __shared__ unsigned bcast[2];
    __shared__ unsigned wsum[2][kSelWaves];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned *keys = keys_all + (size_t)row * N;
    long long *dst = out + (size_t)row * K;
    unsigned need = (unsigned)K;
    const unsigned d1 = radix_pass<11>(keys, N, 0u, 0u, 21, need, hist, bcast, hist1 + (size_t)row * 2048);
    const unsigned d2 = radix_pass<11>(keys, N, 0xFFE00000u, d1 << 21, 10, need, hist, bcast);
    const unsigned p2 = (d1 << 21) | (d2 << 10);
    const unsigned d3 = radix_pass<10>(keys, N, 0xFFFFFC00u, p2, 0, need, hist, bcast);
    const unsigned T = p2 | d3;  // the K-th largest key; `need` of the elements equal to it are taken, in index order
    // ordered output without per-chunk barriers: every wave owns one contiguous 1/16 of the row, counts its winners,
    // the sixteen counts are scanned once, then the wave walks its range again and writes at its running offsets
    const int per_wave = ((N + kSelWaves - 1) / kSelWaves + 255) & ~255;
    const int w0 = wave * per_wave, w1 = min(N, w0 + per_wave);
    const bool vec = ((N & 3) == 0) && ((reinterpret_cast<uintptr_t>(keys) & 15) == 0);  // w0, w1 are multiples of 4 then
    const unsigned long long below = (1ull << lane) - 1ull;
    unsigned cnt_gt = 0, cnt_eq = 0;
    if (vec) {
        // lane l holds elements 4l..4l+3 of a 256-element step: element order = lane order, then component order
        const uint4 *k4 = reinterpret_cast<const uint4 *>(keys);
        constexpr int UF = 2;
        for (int n0 = w0 + 4 * lane; n0 < w1; n0 += 256 * UF) {
            uint4 k[UF];
#pragma unroll
            for (int q = 0; q < UF; ++q) k[q] = n0 + 256 * q < w1 ? k4[(n0 + 256 * q) >> 2] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int q = 0; q < UF; ++q) {
                const bool in = n0 + 256 * q < w1;
                const unsigned c[4] = {k[q].x, k[q].y, k[q].z, k[q].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    cnt_gt += (unsigned)__popcll(__ballot(in && c[e] > T));
                    cnt_eq += (unsigned)__popcll(__ballot(in && c[e] == T));
                }
            }
        }
    } else {
        constexpr int UF = 4;
        for (int n0 = w0 + lane; n0 < w1; n0 += 64 * UF) {
            unsigned k[UF];
#pragma unroll
            for (int q = 0; q < UF; ++q) k[q] = n0 + 64 * q < w1 ? keys[n0 + 64 * q] : 0u;
#pragma unroll
            for (int q = 0; q < UF; ++q) {
                const bool in = n0 + 64 * q < w1;
                cnt_gt += (unsigned)__popcll(__ballot(in && k[q] > T));
                cnt_eq += (unsigned)__popcll(__ballot(in && k[q] == T));
            }
        }
    }
    if (lane == 0) { wsum[0][wave] = cnt_gt; wsum[1][wave] = cnt_eq; }
    __syncthreads();
    unsigned run_gt = 0, run_eq = 0;
    for (int w2 = 0; w2 < wave; ++w2) { run_gt += wsum[0][w2]; run_eq += wsum[1][w2]; }
    if (vec) {
        const uint4 *k4 = reinterpret_cast<const uint4 *>(keys);
        constexpr int UF = 2;
        for (int n0 = w0 + 4 * lane; n0 < w1; n0 += 256 * UF) {
            uint4 k[UF];
#pragma unroll
            for (int q = 0; q < UF; ++q) k[q] = n0 + 256 * q < w1 ? k4[(n0 + 256 * q) >> 2] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int q = 0; q < UF; ++q) {
                const int n = n0 + 256 * q;
                const bool in = n < w1;
                const unsigned c[4] = {k[q].x, k[q].y, k[q].z, k[q].w};
                bool gt[4], eq[4];
                unsigned long long bg[4], be[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    gt[e] = in && c[e] > T;
                    eq[e] = in && c[e] == T;
                    bg[e] = __ballot(gt[e]);
                    be[e] = __ballot(eq[e]);
                }
                // winners before this lane's first element: all four components of the lower lanes
                unsigned pre_gt = run_gt, pre_eq = run_eq;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pre_gt += (unsigned)__popcll(bg[e] & below);
                    pre_eq += (unsigned)__popcll(be[e] & below);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (gt[e] || (eq[e] && pre_eq < need)) dst[pre_gt + (pre_eq < need ? pre_eq : need)] = n + e;
                    pre_gt += gt[e] ? 1u : 0u;
                    pre_eq += eq[e] ? 1u : 0u;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    run_gt += (unsigned)__popcll(bg[e]);
                    run_eq += (unsigned)__popcll(be[e]);
                }
            }
        }
    } else {
        constexpr int UF = 4;
        for (int n0 = w0 + lane; n0 < w1; n0 += 64 * UF) {
            unsigned k[UF];
#pragma unroll
            for (int q = 0; q < UF; ++q) k[q] = n0 + 64 * q < w1 ? keys[n0 + 64 * q] : 0u;
#pragma unroll
            for (int q = 0; q < UF; ++q) {
                const int n = n0 + 64 * q;
                const bool in = n < w1;
                const bool gt = in && k[q] > T, eq = in && k[q] == T;
                const unsigned long long bg = __ballot(gt), be = __ballot(eq);
                const unsigned pre_gt = run_gt + (unsigned)__popcll(bg & below), pre_eq = run_eq + (unsigned)__popcll(be & below);
                if (gt || (eq && pre_eq < need)) dst[pre_gt + (pre_eq < need ? pre_eq : need)] = n;
                run_gt += (unsigned)__popcll(bg);
                run_eq += (unsigned)__popcll(be);
            }
        }
    }
}

// GFNet.sample's gathers (network.py:403-404, 414): om[b][i] = m[b][idx[b][i]] (rows of 4 floats), oc[b][i] = c[b][idx[b][i]]
// with the certainty threshold applied on the way (c > one_above -> 1; +inf = off).
__global__ __launch_bounds__(256) void gather_matches_kernel(const float4 *__restrict__ m, const float *__restrict__ c,
                                                             const long long *__restrict__ idx, float4 *__restrict__ om,
                                                             float *__restrict__ oc, int N, int K, long total, float one_above) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long b = i / K;
    const long src = b * N + idx[i];
    om[i] = m[src];
    const float cv = c[src];
    oc[i] = cv > one_above ? 1.f : cv;
}

}  // namespace

GFN_EXPORT int gfn_gather_matches(const float *matches, const float *certainty, const int64_t *idx, float *out_matches,
                                  float *out_certainty, int Bt, int N, int K, float one_above, gfn_stream_t stream) {
    if (!matches || !certainty || !idx || !out_matches || !out_certainty || Bt < 0 || N <= 0 || K < 0 ||
        (((uintptr_t)matches | (uintptr_t)out_matches) & 15))
        return gfn::fail(GFN_ERR_INVALID_ARG, "gather_matches: bad argument");
    const long total = (long)Bt * K;
    if (total == 0) return GFN_OK;
    hipLaunchKernelGGL(gather_matches_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(matches), certainty, reinterpret_cast<const long long *>(idx),
                       reinterpret_cast<float4 *>(out_matches), out_certainty, N, K, total, one_above);
    return gfn::check_launch("gather_matches_kernel");
}

GFN_EXPORT int gfn_sample_without_replacement(const float *weights, int64_t row_stride, int64_t *out, int *scratch, int Bt, int N, int K,
                                              uint64_t seed, float one_above, gfn_stream_t stream) {
    if (!weights || !out || !scratch || Bt < 0 || N <= 0 || K <= 0 || row_stride < N)
        return gfn::fail(GFN_ERR_INVALID_ARG, "sample_without_replacement: bad argument");
    if (K > N) return gfn::fail(GFN_ERR_INVALID_ARG, "sample_without_replacement: cannot draw %d of %d without replacement", K, N);
    if ((long)Bt * ((N + kKeysPerBlock - 1) / kKeysPerBlock) > 0x7fffffffL) return gfn::fail(GFN_ERR_INVALID_ARG, "sample_without_replacement: too large");
    if (Bt == 0) return GFN_OK;
    hipStream_t s = (hipStream_t)stream;
    unsigned *keys = reinterpret_cast<unsigned *>(scratch);
    unsigned *hist1 = keys + (size_t)Bt * N;
    if (hipMemsetAsync(hist1, 0, sizeof(unsigned) * 2048 * (size_t)Bt, s) != hipSuccess)
        return gfn::fail(GFN_ERR_LAUNCH, "sample_without_replacement: memset failed");
    const int bpr = (N + kKeysPerBlock - 1) / kKeysPerBlock;
    hipLaunchKernelGGL(race_keys_kernel, dim3((unsigned)(Bt * bpr)), dim3(256), 0, s, weights, (long)row_stride, keys, hist1, N, bpr, seed, one_above);
    if (int e = gfn::check_launch("race_keys_kernel")) return e;
    hipLaunchKernelGGL(race_select_kernel, dim3(Bt), dim3(kSelThreads), 0, s, keys, hist1, reinterpret_cast<long long *>(out), N, K);
    return gfn::check_launch("race_select_kernel");
}
