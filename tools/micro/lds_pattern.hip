// Microbenchmark: LDS-pipe cycles per wave-instruction for a given 64-lane address pattern (8 waves issue back to back,
// so the LDS pipe, not one wave's issue rate, is the limit).
// usage: lds_pattern   (runs the built-in list of patterns from the conv-block kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#include <functional>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
enum Op { R32, R64, R128, W32, W128 };
template <int OP>
__global__ void k(const int *offs, long long *out, float *sink) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i;
    __syncthreads();
    const unsigned a = (unsigned)(size_t)lds + offs[threadIdx.x & 63] * 4;  // LDS byte address (every wave the same pattern)
    f4 v4 = {1, 2, 3, 4}; f2 v2; float v1;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        if (OP == R32) asm volatile("ds_read_b32 %0, %1" : "=v"(v1) : "v"(a));
        if (OP == R64) asm volatile("ds_read_b64 %0, %1" : "=v"(v2) : "v"(a));
        if (OP == R128) asm volatile("ds_read_b128 %0, %1" : "=v"(v4) : "v"(a));
        if (OP == W32) asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v4.x));
        if (OP == W128) asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(v4));
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    __syncthreads();
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) *out = t1 - t0;
    sink[threadIdx.x] = v4.x + v2.x + v1;
}
int main() {
    int *d; long long *o; float *sink;
    hipMalloc(&d, 256); hipMalloc(&o, 8); hipMalloc(&sink, 4096);
    struct P { std::string name; Op op; std::function<int(int)> f; };
    std::vector<P> ps = {
        {"r128 linear (ideal)", R128, [](int l) { return 4 * l; }},
        {"w128 linear (ideal)", W128, [](int l) { return 4 * l; }},
        {"w32 linear (ideal)", W32, [](int l) { return l; }},
        {"r64 all lanes one address", R64, [](int l) { return 0; }},
        {"r32 all lanes one address", R32, [](int l) { return 0; }},
        {"commit write alternated first", W128, [](int l) { int q = l % 10; return (l / 10) * 84 + 8 * q + (((q >> 2) & 1) ? 4 : 0); }},
        {"Bs16 write new (b128 linear per row)", W128, [](int l) { int dg = l & 31, dp = l >> 5; return dp * 256 + 2 * (dg / 8) * 32 + 4 * (dg % 8); }},
        {"dw read NB1 TW32 permuted lanes", R128, [](int l) { int l32 = l & 31; int lp = ((0x73261540u >> (4 * (l32 >> 2))) & 7) * 4 + (l32 & 3); int p = l >> 5; return p * 672 + (lp / 8) * 84 + 8 * (lp % 8) + 4; }},
        {"dw read NB2 TW32 shifted permuted", R128, [](int l) { int l32 = l & 31; int lp = ((0x73261540u >> (4 * (l32 >> 2))) & 7) * 4 + (l32 & 3); int p = l >> 5; int rp = lp / 8; return p * 1008 + 2 * rp * 84 + 8 * (lp % 8) + 4 + 4 * (rp & 1); }},
        {"dw read NS2 TW32 permuted", R128, [](int l) { int l32 = l & 31; int lp = ((0x73261540u >> (4 * (l32 >> 2))) & 7) * 4 + (l32 & 3) + (l & 32); return (lp / 16) * 84 + 4 * (lp % 16) + 4; }},
        {"r128 stride 8 dw", R128, [](int l) { return 8 * l; }},
        {"r32 linear", R32, [](int l) { return l; }},
        {"r64 broadcast 2 addrs", R64, [](int l) { return (l / 32) * 64; }},
        {"r64 linear", R64, [](int l) { return 2 * l; }},
        // depthwise halo reads, NB=1 CPT=4 TW=32: lane -> row l/8 (pitch 84), cell group l%8 (8 dw), +4
        {"dw read NB1 TW32", R128, [](int l) { int dg = l & 31, p = l >> 5; return p * 672 + (dg / 8) * 84 + 8 * (dg % 8) + 4; }},
        // NB=2 without / with the row-pair shift: rows 2*(dg/8)
        {"dw read NB2 TW32 plain", R128, [](int l) { int dg = l & 31, p = l >> 5; return p * 1008 + 2 * (dg / 8) * 84 + 8 * (dg % 8) + 4; }},
        {"dw read NB2 TW32 shifted", R128, [](int l) { int dg = l & 31, p = l >> 5; int rp = dg / 8; return p * 1008 + 2 * rp * 84 + 8 * (dg % 8) + 4 + 4 * (rp & 1); }},
        // NS=2 CPT=2: lane -> row l/16, cells 2*(l%16): 4 dw stride
        {"dw read NS2 TW32", R128, [](int l) { return (l / 16) * 84 + 4 * (l % 16) + 4; }},
        // commit: slot e = lane: q = e%10, hr = e/10 -> hr*84 + 8q (lo), +4 (hi)
        {"commit write lo", W128, [](int l) { return (l / 10) * 84 + 8 * (l % 10); }},
        {"commit write hi", W128, [](int l) { return (l / 10) * 84 + 8 * (l % 10) + 4; }},
        // fp16 B tile writes: ((kg*BN + cell)*4 + (dp&3)) dwords, cell = row*32 + 4*cg (+j), lanes: dp = l/32, dg = l%32
        {"Bs16 write NB2", W32, [](int l) { int dg = l & 31, dp = l >> 5; int cell = 2 * (dg / 8) * 32 + 4 * (dg % 8); return cell * 4 + dp; }},
        // fp32 B tile writes b128: (dp*BN + cell)*2 floats
        {"Bs32 write NB1", W128, [](int l) { int dg = l & 31, dp = l >> 5; int cell = (dg / 8) * 32 + 4 * (dg % 8); return (dp * 128 + cell) * 2; }},
        // matrix operand reads
        {"A operand r128 (col consecutive, kh*BMS)", R128, [](int l) { return ((l >> 5) * 224 + (l & 31)) * 4; }},
        {"B16 operand r128", R128, [](int l) { return ((l >> 5) * 256 + (l & 31)) * 4; }},
        {"B32 operand r32 pair layout", R32, [](int l) { return (l & 31) * 2 + (l >> 5); }},
    };
    for (auto &p : ps) {
        int h[64];
        for (int l = 0; l < 64; ++l) h[l] = p.f(l);
        hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
        long long c = 0;
        for (int rep = 0; rep < 2; ++rep) {
            switch (p.op) {
                case R32: hipLaunchKernelGGL(k<R32>, 1, 512, 0, 0, d, o, sink); break;
                case R64: hipLaunchKernelGGL(k<R64>, 1, 512, 0, 0, d, o, sink); break;
                case R128: hipLaunchKernelGGL(k<R128>, 1, 512, 0, 0, d, o, sink); break;
                case W32: hipLaunchKernelGGL(k<W32>, 1, 512, 0, 0, d, o, sink); break;
                case W128: hipLaunchKernelGGL(k<W128>, 1, 512, 0, 0, d, o, sink); break;
            }
            hipDeviceSynchronize();
            hipMemcpy(&c, o, 8, hipMemcpyDeviceToHost);
        }
        printf("%-48s %6.1f cycles/instr\n", p.name.c_str(), c / 512.0);
    }
    return 0;
}
