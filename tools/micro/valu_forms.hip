// Round 6: what does a wave64 fp32 FMA cost to ISSUE on gfx950, by operand form and by waves per SIMD?
// (tools/micro/pkfma.hip measured 3.25-3.5 "cycles at 2.4 GHz" for v_fmac_f32 at 2-4 waves per SIMD, MI355X_MICROARCH.md says 2.)
// Cycles are counted IN the kernel with s_memtime (shader clock), so the figure does not depend on a guessed frequency; the effective clock is
// printed beside it (s_memtime ticks per wall-clock microsecond).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/valu_forms.hip -o tools/micro/valu_forms && tools/micro/valu_forms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

enum Form { FMAC_VVV, FMAC_SV, FMA_BANKS, FMA_SAMEBANK, MUL_VV, ADD_VV, MOV, PK_FMA, FMAC_DEP, FMAC_VVV_X2, FMA_LIT, PK_FMA_S, NFORMS };
static const char *kNames[NFORMS] = {
    "v_fmac_f32 acc_i, va, vb          (16 independent accumulators, 3 VGPR reads)",
    "v_fmac_f32 acc_i, s, vb           (scalar multiplicand: 2 VGPR reads)",
    "v_fma_f32  acc_i, va_i, vb_i, acc (operands spread over the 4 VGPR banks)",
    "v_fma_f32  acc_i, va, vb, acc     (all three sources in ONE bank)",
    "v_mul_f32  d_i, va, vb            (2 reads)",
    "v_add_f32  d_i, va, d_i           (2 reads)",
    "v_mov_b32  d_i, va                (1 read)",
    "v_pk_fma_f32 acc2_i, a2, b2, acc2 (8 independent pairs = 16 FMAs)",
    "v_fmac_f32 acc, va, vb            (ONE accumulator: dependent chain)",
    "v_fmac_f32 acc_i, va_i, vb        (16 accumulators, 16 different multiplicands: the D-stage's form)",
    "v_fmac_f32 acc_i, 0x3f8ccccd, vb  (literal multiplicand)",
    "v_pk_fma_f32 acc2_i, s2, b2, acc2 (scalar pair multiplicand)",
};

template <int FORM>
__global__ __launch_bounds__(256) void rate(float *out, long long *cyc, int iters, float sa) {
    float acc[16], va[16];
    for (int i = 0; i < 16; ++i) { acc[i] = threadIdx.x * 1e-4f * i; va[i] = 1.f + threadIdx.x * 1e-5f * (i + 1); }
    float a = threadIdx.x * 1e-3f + 0.5f, b = 1.f + blockIdx.x * 1e-6f;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc2[8], a2 = {a, a + 1e-3f}, b2 = {b, b};
    for (int i = 0; i < 8; ++i) acc2[i] = f32x2{acc[2 * i], acc[2 * i + 1]};
    float s = __builtin_amdgcn_readfirstlane(sa);
    f32x2 s2 = {s, s};
    // explicit registers for the bank forms: v40..v55 acc, v56.. a, v72.. b.  Bank = register number mod 4.
    long long t0 = 0, t1 = 0;
    if constexpr (FORM == FMA_BANKS || FORM == FMA_SAMEBANK) {
        // acc in v[56+4i] (bank 0); a in v125 (bank 1) / v124 (bank 0); b in v126 (bank 2) / v120 (bank 0): < 128 VGPRs, 4 waves per SIMD fit
        asm volatile("v_mov_b32 v124, %0\n\tv_mov_b32 v125, %0\n\tv_mov_b32 v126, %1\n\tv_mov_b32 v120, %1" :: "v"(a), "v"(b) : "v124", "v125", "v126", "v120");
#define INIT(i) asm volatile("v_mov_b32 v%c0, 0" :: "i"(56 + 4 * i) : "v56", "v60", "v64", "v68", "v72", "v76", "v80", "v84", "v88", "v92", "v96", "v100", "v104", "v108", "v112", "v116");
        REP16(INIT)
#undef INIT
    }
    __builtin_amdgcn_s_barrier();
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (FORM == FMAC_VVV) {
#define X(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
            REP16(X) REP16(X)
#undef X
        } else if constexpr (FORM == FMAC_SV) {
#define X(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "s"(s), "v"(b));
            REP16(X) REP16(X)
#undef X
        } else if constexpr (FORM == FMA_BANKS) {
#define X(i) asm volatile("v_fma_f32 v%c0, v125, v126, v%c0" :: "i"(56 + 4 * i));
            REP16(X) REP16(X)
#undef X
        } else if constexpr (FORM == FMA_SAMEBANK) {
#define X(i) asm volatile("v_fma_f32 v%c0, v124, v120, v%c0" :: "i"(56 + 4 * i));
            REP16(X) REP16(X)
#undef X
        } else if constexpr (FORM == MUL_VV) {
#define X(i) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(acc[i]) : "v"(a), "v"(b));
            REP16(X) REP16(X)
#undef X
        } else if constexpr (FORM == ADD_VV) {
#define X(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(acc[i]) : "v"(a));
            REP16(X) REP16(X)
#undef X
        } else if constexpr (FORM == MOV) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(acc[i]) : "v"(a));
            REP16(X) REP16(X)
#undef X
        } else if constexpr (FORM == PK_FMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc2[i & 7]) : "v"(a2), "v"(b2));
            REP16(X)
#undef X
        } else if constexpr (FORM == PK_FMA_S) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc2[i & 7]) : "s"(s2), "v"(b2));
            REP16(X)
#undef X
        } else if constexpr (FORM == FMAC_DEP) {
#define X(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[0]) : "v"(a), "v"(b));
            REP16(X) REP16(X)
#undef X
        } else if constexpr (FORM == FMAC_VVV_X2) {
#define X(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(va[i]), "v"(b));
            REP16(X) REP16(X)
#undef X
        } else if constexpr (FORM == FMA_LIT) {
#define X(i) asm volatile("v_fmac_f32 %0, 0x3f8ccccd, %1" : "+v"(acc[i]) : "v"(b));
            REP16(X) REP16(X)
#undef X
        }
    }
    t1 = __builtin_readcyclecounter();
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += acc[i] + va[i];
    for (int i = 0; i < 8; ++i) r += acc2[i][0] + acc2[i][1];
    if constexpr (FORM == FMA_BANKS || FORM == FMA_SAMEBANK) {
        float x;
        asm volatile("v_mov_b32 %0, v56" : "=v"(x));
        r += x;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int FORM>
void run(float *out, long long *cyc, std::vector<long long> &h, int form_fmas) {
    const int iters = 4000;
    printf("%s\n", kNames[FORM]);
    for (int w : {1, 2, 4, 8}) {
        const int grid = 256 * w;  // 256-thread blocks = one wave per SIMD each; w blocks per CU
        hipLaunchKernelGGL(rate<FORM>, dim3(grid), dim3(256), 0, 0, out, cyc, 50, 1.1f);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate<FORM>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.1f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), cyc, grid * 4 * sizeof(long long), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.begin() + grid * 4);
        const double med = (double)h[grid * 2];
        const double insts = (double)iters * (form_fmas == 16 && (FORM == PK_FMA || FORM == PK_FMA_S) ? 16 : 32);
        // per SIMD: w waves interleave; a wave's own cycles per instruction / w = SIMD cycles per wave-instruction
        printf("   %d waves/SIMD: %7.2f wave-cycles per instruction -> %5.2f SIMD cycles per wave-instruction (%5.2f per 64 FMAs); clock %.2f GHz (kernel %.3f ms)\n",
               w, med / insts, med / insts / w, med / insts / w / ((FORM == PK_FMA || FORM == PK_FMA_S) ? 2 : 1), med / (ms * 1e6) , ms);
    }
}

int main() {
    float *out;
    long long *cyc;
    hipMalloc(&out, 1 << 24);
    hipMalloc(&cyc, 1 << 20);
    std::vector<long long> h(1 << 17);
    run<FMAC_VVV>(out, cyc, h, 32);
    run<FMAC_VVV_X2>(out, cyc, h, 32);
    run<FMAC_SV>(out, cyc, h, 32);
    run<FMA_LIT>(out, cyc, h, 32);
    run<FMA_BANKS>(out, cyc, h, 32);
    run<FMA_SAMEBANK>(out, cyc, h, 32);
    run<FMAC_DEP>(out, cyc, h, 32);
    run<MUL_VV>(out, cyc, h, 32);
    run<ADD_VV>(out, cyc, h, 32);
    run<MOV>(out, cyc, h, 32);
    run<PK_FMA>(out, cyc, h, 16);
    run<PK_FMA_S>(out, cyc, h, 16);
    return 0;
}
