// Microbenchmark: v_mfma_f32_4x4x1_16B_f32 -- lane layout of A / B / D and issue rate (round 4: a D-stage candidate for the local
// correlation: 16 blocks of (4 pixels) x (4 cells) x (1 channel), exact fp32 fma per instruction).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma4x4.hip -o tools/micro/mfma4x4 && tools/micro/mfma4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout(float *out) {
    const int lane = threadIdx.x;
    // A[lane] = 100 + lane, B[lane] = 1000 * (lane + 1): D = A * B tells which (a-lane, b-lane) pair every output element used
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(100 + lane), (float)(lane + 1), acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = acc[i];
}

template <int NACC>
__global__ __launch_bounds__(256) void rate(float *out, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
        a += 1e-6f;
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// exactness: one instruction == one fmaf per output element (compare with the host's fmaf on awkward operands)
__global__ void exact(const float *a, const float *b, const float *c, float *out) {
    const int lane = threadIdx.x;
    f32x4 acc = {c[lane * 4 + 0], c[lane * 4 + 1], c[lane * 4 + 2], c[lane * 4 + 3]};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[lane], b[lane], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = acc[i];
}

int main() {
    float *out;
    hipMalloc(&out, 1 << 24);
    layout<<<1, 64>>>(out);
    float h[256];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    // decode: D = (100 + la) * (lb + 1)
    int ok = 1;
    for (int lane = 0; lane < 64; ++lane)
        for (int i = 0; i < 4; ++i) {
            const float v = h[lane * 4 + i];
            int fa = -1, fb = -1;
            for (int la = 0; la < 64 && fa < 0; ++la)
                for (int lb = 0; lb < 64; ++lb)
                    if (v == (float)(100 + la) * (float)(lb + 1)) { fa = la; fb = lb; break; }
            const int ea = (lane / 4) * 4 + i, eb = lane;  // expected: A from lane 4*block + i (row i), B from this lane (column j = lane % 4)
            if (fa != ea || fb != eb) ok = 0;
            if (lane < 8) printf("D[lane %d][reg %d] = A[lane %d] * B[lane %d]\n", lane, i, fa, fb);
        }
    printf("layout D[lane = 4 b + j][reg i] = A[lane 4 b + i] * B[lane 4 b + j]: %s\n", ok ? "CONFIRMED" : "NOT as expected");

    // exactness
    float ha[64], hb[64], hc[256], hd[256], *da, *db, *dc;
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 8) - (1 << 23)) / (float)(1 << 20); };
    for (int i = 0; i < 64; ++i) { ha[i] = rnd() * 1.0000001f; hb[i] = rnd() / 3.f; }
    for (int i = 0; i < 256; ++i) hc[i] = rnd() * 1e-3f;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dc, 1024);
    hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice); hipMemcpy(dc, hc, 1024, hipMemcpyHostToDevice);
    exact<<<1, 64>>>(da, db, dc, out);
    hipMemcpy(hd, out, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int i = 0; i < 4; ++i)
            if (hd[lane * 4 + i] != fmaf(ha[(lane / 4) * 4 + i], hb[lane], hc[lane * 4 + i])) ++bad;
    printf("exactness: %d of 256 outputs differ from fmaf(a, b, c)\n", bad);

    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 8, iters = 20000;
    auto time_it = [&](auto kern, int nacc, const char *name) {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double inst = (double)grid * 4 * iters * nacc;  // wave-instructions
        // 256 CUs x 4 SIMDs; 8 waves per SIMD resident (2048 blocks of 4 waves over 1024 SIMDs x 8)
        printf("%s: %.3f ms, %.2f G wave-instr/s = %.1f T mac/s; at 2.4 GHz: %.2f cycles per instruction per SIMD\n", name, ms, inst / ms / 1e6,
               inst * 256 / ms / 1e9, 2.4e9 * 1024 / (inst / (ms * 1e-3)));
    };
    time_it(rate<1>, 1, "1 accumulator ");
    time_it(rate<4>, 4, "4 accumulators");
    time_it(rate<12>, 12, "12 accumulators");
    return 0;
}
