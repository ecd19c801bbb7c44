// Microbenchmark: sustained v_mfma_f32_32x32x2_f32 rate on every CU (registers only), and the shader clock.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters, long long *cycles) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    long long t0 = __builtin_readcyclecounter();  // s_memtime
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}
int main() {
    float *out; long long *cyc;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu) {
        const int grid = 256 * blocks_per_cu, iters = 20000;
        hipLaunchKernelGGL(mfma_loop<7>, dim3(grid), dim3(256), 0, 0, out, 100, cyc);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop<7>, dim3(grid), dim3(256), 0, 0, out, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        double flops = (double)grid * 4 * iters * 7 * 32 * 32 * 2 * 2;
        printf("blocks/CU %d: %.3f ms, %.1f TFLOP/s; wave0 s_memtime ticks %lld (%.1f per MFMA) -> tick rate %.0f MHz\n", blocks_per_cu, ms,
               flops / ms / 1e9, c, (double)c / (iters * 7.0), c / (ms * 1e3));
    }
    return 0;
}
