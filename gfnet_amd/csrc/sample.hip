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

__global__ __launch_bounds__(256) void race_keys_kernel(const float *__restrict__ w, long w_rs, unsigned *__restrict__ keys, int Bt, int N,
                                                        uint64_t seed) {
    const long total = (long)Bt * N;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int row = (int)(idx / N), i = (int)(idx - (long)row * N);
        const float wi = w[(size_t)row * w_rs + i];
        const uint64_t h = splitmix64(seed ^ splitmix64(((uint64_t)(unsigned)row << 32) | (unsigned)i));
        const float u = ((float)(unsigned)(h >> 40) + 1.0f) * 5.9604644775390625e-08f;  // (0, 1], 24 bits
        const float e = -__logf(u);                                                      // Exp(1); 0 only for u == 1
        float key = wi > 0.f ? wi / fmaxf(e, 1e-30f) : 0.f;
        key = key < 3.0e38f ? key : 3.0e38f;
        keys[idx] = __float_as_uint(key);  // non-negative floats order like their bit patterns
    }
}

constexpr int kSelThreads = 1024;
constexpr int kSelWaves = kSelThreads / 64;

// Find, among the elements whose key matches `prefix` in the bits above `shift + bits`, the digit (bits wide, at `shift`)
// that contains the need-th largest; returns it and lowers `need` by the elements in larger digits.  All threads call it.
template <int BITS>
__device__ unsigned radix_pass(const unsigned *__restrict__ keys, int N, unsigned prefix_mask, unsigned prefix, int shift, unsigned &need,
                               unsigned *hist, unsigned *bcast) {
    constexpr int BINS = 1 << BITS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < BINS; e += kSelThreads) hist[e] = 0;
    __syncthreads();
    for (int n = tid; n < N; n += kSelThreads) {
        const unsigned k = keys[n];
        if ((k & prefix_mask) == prefix) atomicAdd(&hist[(k >> shift) & (BINS - 1)], 1u);
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

__global__ __launch_bounds__(kSelThreads) void race_select_kernel(const unsigned *__restrict__ keys_all, long long *__restrict__ out, int N,
                                                                  int K) {
    __shared__ unsigned hist[2048];
    __shared__ unsigned bcast[2];
    __shared__ unsigned wsum[2][kSelWaves];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned *keys = keys_all + (size_t)row * N;
    long long *dst = out + (size_t)row * K;
    unsigned need = (unsigned)K;
    const unsigned d1 = radix_pass<11>(keys, N, 0u, 0u, 21, need, hist, bcast);
    const unsigned d2 = radix_pass<11>(keys, N, 0xFFE00000u, d1 << 21, 10, need, hist, bcast);
    const unsigned p2 = (d1 << 21) | (d2 << 10);
    const unsigned d3 = radix_pass<10>(keys, N, 0xFFFFFC00u, p2, 0, need, hist, bcast);
    const unsigned T = p2 | d3;  // the K-th largest key; `need` of the elements equal to it are taken, in index order
    unsigned run_gt = 0, run_eq = 0;
    for (int n0 = 0; n0 < N; n0 += kSelThreads) {
        const int n = n0 + tid;
        const unsigned k = n < N ? keys[n] : 0u;
        const bool gt = n < N && k > T, eq = n < N && k == T;
        const unsigned long long bg = __ballot(gt), be = __ballot(eq);
        if (lane == 0) { wsum[0][wave] = (unsigned)__popcll(bg); wsum[1][wave] = (unsigned)__popcll(be); }
        __syncthreads();
        unsigned pre_gt = run_gt, pre_eq = run_eq, tot_gt = 0, tot_eq = 0;
#pragma unroll
        for (int w2 = 0; w2 < kSelWaves; ++w2) {
            const unsigned g = wsum[0][w2], e = wsum[1][w2];
            if (w2 < wave) { pre_gt += g; pre_eq += e; }
            tot_gt += g; tot_eq += e;
        }
        const unsigned long long below = (1ull << lane) - 1ull;
        pre_gt += (unsigned)__popcll(bg & below);
        pre_eq += (unsigned)__popcll(be & below);
        if (gt || (eq && pre_eq < need)) dst[pre_gt + (pre_eq < need ? pre_eq : need)] = n;
        run_gt += tot_gt;
        run_eq += tot_eq;
        __syncthreads();
    }
}

}  // namespace

GFN_EXPORT int gfn_sample_without_replacement(const float *weights, int64_t row_stride, int64_t *out, int *scratch, int Bt, int N, int K,
                                              uint64_t seed, gfn_stream_t stream) {
    if (!weights || !out || !scratch || Bt < 0 || N <= 0 || K <= 0 || row_stride < N)
        return gfn::fail(GFN_ERR_INVALID_ARG, "sample_without_replacement: bad argument");
    if (K > N) return gfn::fail(GFN_ERR_INVALID_ARG, "sample_without_replacement: cannot draw %d of %d without replacement", K, N);
    if (Bt == 0) return GFN_OK;
    hipStream_t s = (hipStream_t)stream;
    unsigned *keys = reinterpret_cast<unsigned *>(scratch);
    const long total = (long)Bt * N;
    const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(race_keys_kernel, dim3(grid), dim3(256), 0, s, weights, (long)row_stride, keys, Bt, N, seed);
    if (int e = gfn::check_launch("race_keys_kernel")) return e;
    hipLaunchKernelGGL(race_select_kernel, dim3(Bt), dim3(kSelThreads), 0, s, keys, reinterpret_cast<long long *>(out), N, K);
    return gfn::check_launch("race_select_kernel");
}
