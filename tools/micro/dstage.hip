// Microbenchmark for the round-2 local-correlation D-stage: one cell per wave pass, lane = (patch row, aligned
// pixel pair), f1 region staged channel-planar in LDS, one ds_read_b64 + one v_pk_fma_f32 per channel, the cell's
// f0 value wave-uniform (SGPR pair operand with op_sel broadcast, or a VGPR filled by an LDS broadcast read).
// Reports (a) whether v_pk_fma_f32 with an SGPR-pair operand + op_sel computes what it should on gfx950 and
// (b) CU-cycles per cell for the modes, at 1..4 workgroups of 256 threads per CU, plus bare FMA issue rates.
//   build: hipcc -O3 --offload-arch=gfx950 tools/micro/dstage.hip -o tools/micro/dstage
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f8 __attribute__((ext_vector_type(8)));

constexpr int PS = 648;       // plane stride (pixels); 2592 B: not a multiple of 512 B and > 2040 B, so the compiler cannot fuse two
                              // reads into ds_read2(st64)_b64 (half the LDS rate per byte)
constexpr int CH = 16;        // channels per staged chunk
constexpr int NCW = 8;        // cells per wave and tile
constexpr int C = 32;

// MODE 0: f0 in SGPRs (s_load_dwordx8 per channel: 8 consecutive cells), v_pk_fma_f32 with SGPR pair + op_sel
// MODE 1: f0 in VGPRs from an LDS broadcast read (f0s[cell][C]), v_pk_fma_f32 with VGPR + op_sel
// MODE 2: f0 in SGPRs, two plain v_fma_f32
// MODE 3: the LDS reads of mode 0 alone (results consumed by an empty asm)
// MODE 4: the packed FMAs of mode 1 alone (operands stay in registers)
template <int MODE>
__global__ __launch_bounds__(256) void dstage(const float *__restrict__ f0, int G2, const float *__restrict__ fill, int pitch,
                                              int tiles, float *out, int lanes_active, int wrap) {
    __shared__ __attribute__((aligned(16))) float S[CH * PS];
    __shared__ __attribute__((aligned(16))) float f0s[32 * (C + 4)];
    for (int i = threadIdx.x; i < CH * PS; i += 256) S[i] = fill[i];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: s_load for f0
    const int y = lane / 6, pr = lane % 6;
    const bool act = lane < lanes_active;
    const unsigned lane_off = act ? (unsigned)((y * pitch + 2 * pr) * 4) : 0u;
    f2 acc[NCW];
    float total = 0.f;
    for (int t = 0; t < tiles; ++t) {
        const int cell0 = (blockIdx.x * tiles + t) * 32 + wave * NCW;  // first of this wave's 8 cells
        if (MODE == 1) {
            __syncthreads();
            for (int e = threadIdx.x; e < 32 * C; e += 256) {
                const int c = e >> 5, cell = e & 31;
                f0s[cell * (C + 4) + c] = f0[(size_t)c * G2 + (blockIdx.x * tiles + t) * 32 + cell];
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NCW; ++i) acc[i] = (f2){0.f, 0.f};
#pragma unroll 1
        for (int c0 = 0; c0 < C; c0 += 8) {
            f8 s[8];
            if (MODE == 0 || MODE == 2) {
#pragma unroll
                for (int k = 0; k < 8; ++k) s[k] = *reinterpret_cast<const f8 *>(f0 + (size_t)(c0 + k) * G2 + (cell0 & wrap));
            }
            const int cl = c0 & (CH - 1);  // channel inside the staged chunk (the bench re-uses the same 16 planes)
#pragma unroll
            for (int i = 0; i < NCW; ++i) {
                const int X0 = 2 * i + (wave & 1), Y0 = wave + (i & 1) + t % 3;
                const unsigned a = lane_off + (unsigned)((Y0 * pitch + (X0 & ~1)) * 4);
                const char *sp = reinterpret_cast<const char *>(S) + a;
                f2 v[8];
                if (MODE != 4) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const f2 *>(sp + (cl + k) * PS * 4);
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = (f2){(float)(lane + k), (float)(i + t)};
                }
                if (MODE == 3) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) asm volatile("" ::"v"(v[k]));
                } else if (MODE == 4) {
                    const f2 p = {(float)c0, (float)i};
#pragma unroll
                    for (int k = 0; k < 8; ++k) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[i]) : "v"(v[k]), "v"(p));
                } else if (MODE == 1) {
                    const f4 *fq = reinterpret_cast<const f4 *>(f0s + (wave * NCW + i) * (C + 4) + c0);
                    const f4 q0 = fq[0], q1 = fq[1];
                    const f2 p[4] = {{q0.x, q0.y}, {q0.z, q0.w}, {q1.x, q1.y}, {q1.z, q1.w}};
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        if (k & 1)
                            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc[i]) : "v"(v[k]), "v"(p[k >> 1]));
                        else
                            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[i]) : "v"(v[k]), "v"(p[k >> 1]));
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const f2 sp2 = {s[k][i & ~1], s[k][i | 1]};
                        if (MODE == 0) {
                            if (i & 1)
                                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc[i]) : "v"(v[k]), "s"(sp2));
                            else
                                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[i]) : "v"(v[k]), "s"(sp2));
                        } else {
                            const float sc = s[k][i];
                            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(v[k].x), "s"(sc));
                            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].y) : "v"(v[k].y), "s"(sc));
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NCW; ++i) total += acc[i].x + 2.f * acc[i].y + (float)i * acc[i].x;
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = act ? total : 0.f;
}

// bare issue-rate loops: N independent accumulators, plain vs packed FMA
template <int PK>
__global__ __launch_bounds__(256) void fma_rate(float *out, int iters, float a, float b) {
    f2 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (f2){(float)threadIdx.x, (float)i};
    const f2 va = {a, a}, vb = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (PK)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(va), "v"(vb));
            else {
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(va.x), "v"(vb.x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].y) : "v"(va.y), "v"(vb.y));
            }
        }
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += acc[i].x + acc[i].y;
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = t;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int M>
static void launch(int nwg, const float *f0, int G2, const float *fill, int pitch, int tiles, float *out, int lanes, int wrap) {
    hipLaunchKernelGGL(dstage<M>, nwg, 256, 0, 0, f0, G2, fill, pitch, tiles, out, lanes, wrap);
}
static void launch_mode(int m, int nwg, const float *f0, int G2, const float *fill, int pitch, int tiles, float *out, int lanes, int wrap) {
    switch (m) {
        case 0: launch<0>(nwg, f0, G2, fill, pitch, tiles, out, lanes, wrap); break;
        case 1: launch<1>(nwg, f0, G2, fill, pitch, tiles, out, lanes, wrap); break;
        case 2: launch<2>(nwg, f0, G2, fill, pitch, tiles, out, lanes, wrap); break;
        case 3: launch<3>(nwg, f0, G2, fill, pitch, tiles, out, lanes, wrap); break;
        case 4: launch<4>(nwg, f0, G2, fill, pitch, tiles, out, lanes, wrap); break;
    }
}

int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const int nwg_max = 256 * 8, tiles = 16, G2 = 1024 * tiles * 32;
    std::vector<float> hf0((size_t)C * G2), hfill(CH * PS);
    unsigned st = 12345;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (auto &v : hf0) v = rnd();
    for (auto &v : hfill) v = rnd();
    float *f0, *fill, *out;
    CK(hipMalloc(&f0, hf0.size() * 4 + 4096)); CK(hipMalloc(&fill, hfill.size() * 4)); CK(hipMalloc(&out, (size_t)nwg_max * 256 * 4));
    CK(hipMemcpy(f0, hf0.data(), hf0.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(fill, hfill.data(), hfill.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int ALL = 0x7fffffff;

    // ---- correctness of the SGPR-operand packed FMA: modes 0, 1, 2 must agree ------------------------------
    std::vector<float> r[3];
    for (int m = 0; m < 3; ++m) {
        CK(hipMemset(out, 0, (size_t)nwg_max * 256 * 4));
        launch_mode(m, 8, f0, G2, fill, 38, 2, out, 60, ALL);
        CK(hipDeviceSynchronize());
        r[m].resize(8 * 256);
        CK(hipMemcpy(r[m].data(), out, 8 * 256 * 4, hipMemcpyDeviceToHost));
    }
    double d01 = 0, d02 = 0, mag = 0;
    for (size_t i = 0; i < r[0].size(); ++i) { d01 = fmax(d01, fabs(r[0][i] - r[1][i])); d02 = fmax(d02, fabs(r[0][i] - r[2][i])); mag = fmax(mag, fabs(r[0][i])); }
    printf("check: max|sgpr-pk - vgpr-pk| = %.3g, max|sgpr-pk - sgpr-plain| = %.3g (max |value| %.3g)\n", d01, d02, mag);

    // ---- D-stage cost ------------------------------------------------------------------------------------
    const char *names[5] = {"pk_fma, f0 SGPR (s_load_dwordx8)", "pk_fma, f0 VGPR (LDS broadcast)", "2 x v_fma, f0 SGPR", "LDS reads only", "pk_fma only"};
    auto run = [&](int m, int pitch, int wgpc, int lanes, int wrap, const char *tag) {
        const int nwg = 256 * wgpc;
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            launch_mode(m, nwg, f0, G2, fill, pitch, tiles, out, lanes, wrap);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        const double cells_per_cu = (double)wgpc * tiles * 32;
        printf("pitch %2d lanes %2d %-34s %-10s %d WG/CU: %7.1f us -> %6.1f cycles@2.4GHz per cell(32ch) per CU\n", pitch, lanes, names[m], tag, wgpc,
               ms * 1e3, ms * 1e-3 / cells_per_cu * 2.4e9);
    };
    for (int m : {3, 4, 1, 0, 2})
        for (int wgpc : {1, 2, 3}) run(m, 38, wgpc, 60, ALL, m == 0 || m == 2 ? "f0 cold" : "");
    for (int m : {0, 2})
        for (int wgpc : {1, 2, 3}) run(m, 38, wgpc, 60, 4095, "f0 warm");   // 4096 cells x 32 ch = 512 KB of f0, re-read by everyone
    for (int pitch : {44, 32, 12}) {
        for (int m : {3, 1}) run(m, pitch, 3, 60, ALL, "");
    }
    for (int m : {3, 1, 0}) run(m, 38, 3, 24, 4095, "r=2 lanes");
    // ---- bare FMA issue rates ------------------------------------------------------------------------------
    for (int pk = 0; pk < 2; ++pk)
        for (int wgpc : {1, 2, 4, 8}) {
            const int iters = 4096;
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                if (pk) hipLaunchKernelGGL(fma_rate<1>, 256 * wgpc, 256, 0, 0, out, iters, 1.0001f, 0.5f);
                else hipLaunchKernelGGL(fma_rate<0>, 256 * wgpc, 256, 0, 0, out, iters, 1.0001f, 0.5f);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            const double fma = (double)256 * wgpc * 256 * iters * 16;
            printf("%s, %d waves/SIMD: %.1f TFLOP/s\n", pk ? "v_pk_fma_f32" : "v_fma_f32   ", wgpc, 2 * fma / (ms * 1e-3) / 1e12);
        }
    return 0;
}
