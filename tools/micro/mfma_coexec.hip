// mfma_coexec.hip -- do v_mfma_f32_16x16x4_f32 and plain VALU work of OTHER waves of the same SIMD overlap? (round 5)
// 4 waves per SIMD (two 512-thread workgroups per CU): MODE 0 all waves issue MFMAs, MODE 1 all waves issue independent v_fma_f32,
// MODE 2 waves 0-3 MFMAs / waves 4-7 v_fma_f32 (one of each per SIMD) (the same per-wave work as in modes 0 and 1).  If the matrix pipe and the VALU overlap
// across waves, mode 2 takes ~max(mode 0, mode 1) / ... of the halves; if the fp32 MFMA holds the vector ALU, ~the sum.
// hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_coexec.hip -o tools/micro/mfma_coexec && tools/micro/mfma_coexec
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

template <int MODE, bool BF16>
__global__ __launch_bounds__(512, 2) void k(float *out, int iters) {
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = MODE == 0 || (MODE == 2 && wave < 4);   // (waves w and w + 4 share a SIMD: one MFMA wave and one VALU wave per SIMD and workgroup)
    const bool do_valu = MODE == 1 || (MODE == 2 && wave >= 4);
    f32x4 acc[8];
    float v[16];
    for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < 16; ++t) v[t] = (float)(threadIdx.x + t);
    const float a = (float)threadIdx.x, b = 1.0009765625f;
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    bf16x8 ha, hb;
    for (int e = 0; e < 8; ++e) { ha[e] = (__bf16)(float)(threadIdx.x + e); hb[e] = (__bf16)1.0f; }
    for (int it = 0; it < iters; ++it) {
        if (do_mfma) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (BF16) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acc[t], 0, 0, 0);
                else acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
            }
        }
        if (do_valu) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < 16; ++t) v[t] = fmaf(v[t], b, a);   // 64 independent-ish FMAs per iteration (16 chains)
        }
    }
    float s = 0.f;
    for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    for (int t = 0; t < 16; ++t) s += v[t];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, bool BF16>
float run(float *out, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<MODE, BF16>), dim3(512), dim3(512), 0, 0, out, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE, BF16>), dim3(512), dim3(512), 0, 0, out, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f;
}

int main() {
    float *out;
    CK(hipMalloc(&out, 512 * 512 * 4));
    const int iters = 2000;
    printf("fp32 MFMA 16x16x4 : all waves MFMA %8.1f us | all waves VALU %8.1f us | waves 0-3 MFMA + 4-7 VALU %8.1f us\n", run<0, false>(out, iters),
           run<1, false>(out, iters), run<2, false>(out, iters));
    printf("bf16 MFMA 16x16x32: all waves MFMA %8.1f us | all waves VALU %8.1f us | waves 0-3 MFMA + 4-7 VALU %8.1f us\n", run<0, true>(out, iters),
           run<1, true>(out, iters), run<2, true>(out, iters));
    printf("(mode 2 does half of mode 0's MFMAs and half of mode 1's FMAs: full overlap -> max(m0, m1) / 2, none -> (m0 + m1) / 2)\n");
    return 0;
}
