// mfma16x16x4.hip -- v_mfma_f32_16x16x4_f32 as the local correlation's D-stage would use it (round 5):
//   (1) operand / result layout and bitwise equality with a sequential fmaf chain over k (the D-stage's channel order);
//   (2) SIMD cycles per instruction with 4 waves per SIMD: MFMAs alone, with one ds_read_b128 per 4 MFMAs, and with VALU work of the
//       same wave between them (what the rest of the tile kernel issues).
// hipcc -O3 --offload-arch=gfx950 tools/micro/mfma16x16x4.hip -o tools/micro/mfma16x16x4 && tools/micro/mfma16x16x4
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

// one wave: A[16][K], B[K][16] row-major in memory, K a multiple of 4, D[16][16]
__global__ void layout_kernel(const float *A, const float *B, float *D, int K) {
    const int l = threadIdx.x, i = l & 15, kk = l >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 4) {
        const float a = A[i * K + k0 + kk];        // lane (i, kk) supplies A[i][k0 + kk]
        const float b = B[(k0 + kk) * 16 + i];     // lane (j = i, kk) supplies B[k0 + kk][j]
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    for (int v = 0; v < 4; ++v) D[(4 * kk + v) * 16 + i] = acc[v];   // lane (j, kk), register v: D[4 kk + v][j]
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void rate_kernel(float *out, long long *cycles, int iters) {
    __shared__ float4 lds[1024];
    const int l = threadIdx.x & 63;
    lds[threadIdx.x] = make_float4((float)l, 1.f, 2.f, 3.f);
    lds[threadIdx.x + 512] = make_float4((float)l, 1.f, 2.f, 3.f);
    __syncthreads();
    f32x4 acc[10];
#pragma unroll
    for (int t = 0; t < 10; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 a = lds[l], b = lds[l + 64];
    float v0 = (float)l, v1 = 1.f, v2 = 2.f, v3 = 3.f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 10; ++t) {
            if (MODE >= 1) b = lds[(l + 64 * t + it) & 1023];
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[t], 0, 0, 0);
            if (MODE == 2) { v0 = fmaf(v0, v1, v2); v1 = fmaf(v1, v2, v3); }
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[t], 0, 0, 0);
            if (MODE == 2) { v2 = fmaf(v2, v3, v0); v3 = fmaf(v3, v0, v1); }
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc[t], 0, 0, 0);
            if (MODE == 2) { v0 = fmaf(v0, v1, v2); v1 = fmaf(v1, v2, v3); }
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc[t], 0, 0, 0);
            if (MODE == 2) { v2 = fmaf(v2, v3, v0); v3 = fmaf(v3, v0, v1); }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = v0 + v1 + v2 + v3;
#pragma unroll
    for (int t = 0; t < 10; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
    // (1) layout + exactness
    const int K = 32;
    std::vector<float> A(16 * K), B(K * 16), D(256), R(256);
    srand(1);
    for (auto &v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto &v : B) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            float acc = 0.f;
            for (int k = 0; k < K; ++k) acc = fmaf(A[i * K + k], B[k * 16 + j], acc);
            R[i * 16 + j] = acc;
        }
    float *dA, *dB, *dD;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dD, 1024));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
    CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
    int diff = 0;
    double maxerr = 0;
    for (int e = 0; e < 256; ++e) { diff += D[e] != R[e]; maxerr = fmax(maxerr, fabs((double)D[e] - R[e])); }
    printf("layout/exactness: %d of 256 outputs differ from the sequential fmaf chain over k (max abs %.3g)\n", diff, maxerr);
    // (2) rates: 512 workgroups of 512 threads (two per CU, 4 waves per SIMD)
    float *out;
    long long *cyc;
    const int nb = 512, iters = 200;
    CK(hipMalloc(&out, (size_t)nb * 512 * 4)); CK(hipMalloc(&cyc, nb * 8));
    std::vector<long long> h(nb);
    auto run = [&](int mode, const char *what) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(nb), dim3(512), 0, 0, out, cyc, iters);
            if (mode == 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(nb), dim3(512), 0, 0, out, cyc, iters);
            if (mode == 2) hipLaunchKernelGGL(rate_kernel<2>, dim3(nb), dim3(512), 0, 0, out, cyc, iters);
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(h.data(), cyc, nb * 8, hipMemcpyDeviceToHost));
        double m = 0;
        for (auto v : h) m += (double)v;
        m /= nb;
        // per SIMD: 4 waves (two workgroups of 8 waves per CU), each iters * 40 MFMAs
        printf("%-58s %8.1f cycles per MFMA per wave, %6.1f per MFMA per SIMD (4 waves)\n", what, m / (iters * 40.0), m / (iters * 40.0) / 4.0);
    };
    run(0, "MFMAs alone (10 accumulators, registers)");
    run(1, "one ds_read_b128 per 4 MFMAs");
    run(2, "ds_read_b128 + 2 dependent v_fma_f32 of the wave per MFMA");
    return 0;
}
