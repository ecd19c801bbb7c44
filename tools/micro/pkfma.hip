// Microbenchmark: issue rate of v_fma_f32 against v_pk_fma_f32 (with a broadcast second operand via op_sel) at 1 / 2 / 4 waves per SIMD,
// and the bits of the packed form (round 4: a candidate for the local correlation's D-stage, two patch positions per instruction).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/pkfma.hip -o tools/micro/pkfma && tools/micro/pkfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NACC>
__global__ __launch_bounds__(256) void rate_fma(float *out, int iters) {
    float acc[2 * NACC];
    for (int i = 0; i < 2 * NACC; ++i) acc[i] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 2 * NACC; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    float s = 0.f;
    for (int i = 0; i < 2 * NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void rate_pk(float *out, int iters) {
    f32x2 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x2{0.f, 0.f};
    f32x2 a = {threadIdx.x * 1e-3f, threadIdx.x * 2e-3f}, b = {blockIdx.x * 1e-3f + 1.f, 2.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// bits: lo = fmaf(a.x, b.x or b.y, c.x), hi = fmaf(a.y, the same b, c.y)
__global__ void exact(const float *a, const float *b, const float *c, float *out) {
    const int t = threadIdx.x;
    f32x2 A = {a[2 * t], a[2 * t + 1]}, B = {b[2 * t], b[2 * t + 1]}, C = {c[2 * t], c[2 * t + 1]}, D = C;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(C) : "v"(A), "v"(B));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(D) : "v"(A), "v"(B));
    out[4 * t] = C[0]; out[4 * t + 1] = C[1]; out[4 * t + 2] = D[0]; out[4 * t + 3] = D[1];
}

int main() {
    float *out;
    hipMalloc(&out, 1 << 24);
    float ha[128], hb[128], hc[128], hd[256], *da, *db, *dc;
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 8) - (1 << 23)) / (float)(1 << 20); };
    for (int i = 0; i < 128; ++i) { ha[i] = rnd() * 1.0000001f; hb[i] = rnd() / 3.f; hc[i] = rnd() * 1e-3f; }
    hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dc, 512);
    hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice); hipMemcpy(dc, hc, 512, hipMemcpyHostToDevice);
    exact<<<1, 64>>>(da, db, dc, out);
    hipMemcpy(hd, out, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 64; ++t) {
        if (hd[4 * t] != fmaf(ha[2 * t], hb[2 * t], hc[2 * t])) ++bad;
        if (hd[4 * t + 1] != fmaf(ha[2 * t + 1], hb[2 * t], hc[2 * t + 1])) ++bad;
        if (hd[4 * t + 2] != fmaf(ha[2 * t], hb[2 * t + 1], hc[2 * t])) ++bad;
        if (hd[4 * t + 3] != fmaf(ha[2 * t + 1], hb[2 * t + 1], hc[2 * t + 1])) ++bad;
    }
    printf("v_pk_fma_f32 with a broadcast operand (op_sel lo / hi): %d of 256 outputs differ from fmaf\n", bad);

    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    auto time_it = [&](auto kern, int nacc, int waves_per_simd, const char *name) {
        const int grid = 256 * waves_per_simd;  // blocks of 4 waves: one wave per SIMD each
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fma = (double)grid * 4 * iters * nacc * 2;  // wave-level scalar-FMA equivalents
        printf("%s, %d waves/SIMD: %.3f ms, %.2f cycles per 64 FMAs per SIMD at 2.4 GHz (%.1f T mac/s)\n", name, waves_per_simd, ms,
               2.4e9 * 1024 / (fma / (ms * 1e-3)), fma * 64 / ms / 1e9);
    };
    for (int w : {1, 2, 4, 8}) {
        time_it(rate_fma<8>, 8, w, "v_fmac_f32   ");
        time_it(rate_pk<8>, 8, w, "v_pk_fma_f32 ");
    }
    return 0;
}
