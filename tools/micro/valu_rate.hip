// Microbenchmark: per-lane multiply-add rates of the VALU forms a depthwise 5x5 on fp16 maps could use (registers only).
//   0 v_fma_f32   1 v_pk_fma_f32   2 v_fma_mix_f32 (fp16 source, fp32 accumulate)   3 v_dot2_f32_f16   4 v_pk_fma_f16   5 v_dot2c_f32_f16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr int NA = 16;
template <int KIND>
__global__ __launch_bounds__(256) void loop(float *out, int iters) {
    float a[NA];
    f32x2 a2[NA];
    f16x2 h2[NA];
    for (int i = 0; i < NA; ++i) a[i] = 0.f, a2[i] = f32x2{0.f, 0.f}, h2[i] = f16x2{(_Float16)0.f, (_Float16)0.f};
    float w = threadIdx.x * 1e-3f + 1e-4f, x = blockIdx.x * 1e-3f + 1e-4f;
    f32x2 w2 = {w, w * 0.5f}, x2 = {x, x + 1.f};
    f16x2 hw = {(_Float16)w, (_Float16)(w * 0.5f)}, hx = {(_Float16)x, (_Float16)(x + 1.f)};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(w), "v"(x));
            if constexpr (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a2[i]) : "v"(w2), "v"(x2));
            if constexpr (KIND == 2) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(hx), "v"(w));
            if constexpr (KIND == 3) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(hw), "v"(hx));
            if constexpr (KIND == 4) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(h2[i]) : "v"(hw), "v"(hx));
            if constexpr (KIND == 5) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(a[i]) : "v"(hw), "v"(hx));
        }
    }
    float s = 0.f;
    for (int i = 0; i < NA; ++i) s += a[i] + a2[i].x + a2[i].y + (float)h2[i].x + (float)h2[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND>
void run(const char *name, int macs, float *out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 8, iters = 20000;
    hipLaunchKernelGGL(loop<KIND>, dim3(grid), dim3(256), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(loop<KIND>, dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst = (double)grid * 256 * iters * NA;
    printf("%-18s %.3f ms  %.2f T lane-instr/s  %.1f T mac/s\n", name, ms, inst / ms / 1e9, inst * macs / ms / 1e9);
}
int main() {
    float *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>("v_fma_f32", 1, out);
    run<1>("v_pk_fma_f32", 2, out);
    run<2>("v_fma_mix_f32", 1, out);
    run<3>("v_dot2_f32_f16", 2, out);
    run<4>("v_pk_fma_f16", 2, out);
    run<5>("v_dot2c_f32_f16", 2, out);
    return 0;
}
