// Round 6: the gate micro-benchmark of VERDICT r5 item 2 -- LDS-DMA (buffer_load_dwordx4 ... lds) staging of a 44 x 19-pixel x 32-channel region
// into channel PLANES, and the D-stage forms a planar stage allows -- measured the way the product kernel runs: 512-thread workgroups, 64 KB of
// stage, two workgroups per CU, 4 096 tiles of 64 cells per launch, 10 x 10-pixel patches on a 1.75-pixel cell lattice.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/lds_dma_planar.hip -o tools/micro/lds_dma_planar && tools/micro/lds_dma_planar
//   (counters: rocprofv3 --pmc SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU -- tools/micro/lds_dma_planar MODE)
// MODE 0: fill only (both 16-channel chunks of a tile, vmcnt(0) + barrier behind each)
// MODE 1: fill + planar D-stage on ALIGNED quads: 16 lanes per cell, a lane owns (patch row, aligned quad) items of the cell's 16-pixel window:
//         40 items = 3 passes, one ds_read_b128 + 4 FMAs per item and channel
// MODE 2: fill + planar D-stage on UNALIGNED reads: a lane owns one patch ROW of a cell (10 lanes of 16 busy), two ds_read_b128 + one ds_read_b64 at
//         the patch's own column (4-byte aligned) + 10 FMAs per channel
// MODE 3: as MODE 2 with the rows of the workgroup's 64 cells dealt densely over its lanes (640 items = 1.25 passes of 512 lanes; timing only)
// Every mode checks its sums against the host for the tiles of workgroup 0 (the fill's layout and the zeros of out-of-region lanes included).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 4-byte aligned: the patch's own column
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))

constexpr int kW = 112, kH = 112, kC = 32, kImages = 64;
constexpr int kRW = 44, kRH = 19, kNQ = kRW / 4;     // region: 44 x 19 pixels = 11 quads per row
constexpr int kPlaneBytes = 4096;                     // LDS bytes per channel plane (1024 floats >= 19 x 48)
#ifndef LDS_PITCH
#define LDS_PITCH 48   // pixels per staged row in LDS: 48 = 12 quads (the 12th filled with zeros by the range check) makes the 4 rows x 4 quads of a
#endif                 // 16-lane read group 16 distinct 16-byte units (12 r + q mod 16); 44 = the region's own width: 640 conflict clocks per wave
constexpr int kLP = LDS_PITCH, kLQ = kLP / 4;
constexpr int kChunk = 16;
constexpr int kStage = kChunk * kPlaneBytes;          // 64 KB
constexpr int kPW = 10;
constexpr unsigned kOOB = 0x7FFFFFF0u;

struct Tile {
    int img, x0, y0;
};
__host__ __device__ inline Tile tile_of(int t) {
    Tile r;
    r.img = t & (kImages - 1);
    r.x0 = 4 * ((t * 7) % 16);          // 0..60, a multiple of 4: + 44 <= 112 for x0 <= 68
    r.y0 = (t * 5) % (kH - kRH + 1);    // 0..93
    return r;
}
// patch origin of cell (i, j) of a tile: a 1.75-pixel lattice with a little shear, inside the region
__host__ __device__ inline void cell_origin(int i, int j, int &X, int &Y) {
    X = (int)(1.75f * j + 0.3f * i);
    Y = (int)(1.75f * i + 0.2f * j);
}
__host__ __device__ inline float f0_of(int cell, int c) { return 0.25f + 0.01f * (float)((cell * 7 + c * 3) % 31); }


// four channels of a patch row through UNALIGNED (4-byte aligned) 16- and 8-byte LDS reads: hipcc splits such loads into ds_read2_b32 pairs, so
// they are written out; one wait for the batch of twelve
#define ROW_BATCH(BASE, C0, FQ, AC)                                                                                                     \
    {                                                                                                                                  \
        f32x4 a0, a1, b0, b1, c0_, c1_, d0, d1;                                                                                        \
        f32x2 a2, b2, c2_, d2;                                                                                                         \
        const unsigned ad = (BASE) + (C0) * kPlaneBytes;                                                                               \
        asm volatile("ds_read_b128 %0, %12\n\tds_read_b128 %1, %12 offset:16\n\tds_read_b64 %2, %12 offset:32\n\t"                       \
                     "ds_read_b128 %3, %12 offset:4096\n\tds_read_b128 %4, %12 offset:4112\n\tds_read_b64 %5, %12 offset:4128\n\t"         \
                     "ds_read_b128 %6, %12 offset:8192\n\tds_read_b128 %7, %12 offset:8208\n\tds_read_b64 %8, %12 offset:8224\n\t"         \
                     "ds_read_b128 %9, %12 offset:12288\n\tds_read_b128 %10, %12 offset:12304\n\tds_read_b64 %11, %12 offset:12320\n\t"    \
                     "s_waitcnt lgkmcnt(0)"                                                                                            \
                     : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(c0_), "=&v"(c1_), "=&v"(c2_), "=&v"(d0),  \
                       "=&v"(d1), "=&v"(d2)                                                                                            \
                     : "v"(ad)                                                                                                         \
                     : "memory");                                                                                                      \
        const f32x4 *q0[4] = {&a0, &b0, &c0_, &d0}, *q1[4] = {&a1, &b1, &c1_, &d1};                                                    \
        const f32x2 *q2[4] = {&a2, &b2, &c2_, &d2};                                                                                    \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                                                \
            const float f = (FQ)[(C0) + u];                                                                                            \
            (AC)[0] = fmaf(f, (*q0[u])[0], (AC)[0]); (AC)[1] = fmaf(f, (*q0[u])[1], (AC)[1]);                                           \
            (AC)[2] = fmaf(f, (*q0[u])[2], (AC)[2]); (AC)[3] = fmaf(f, (*q0[u])[3], (AC)[3]);                                           \
            (AC)[4] = fmaf(f, (*q1[u])[0], (AC)[4]); (AC)[5] = fmaf(f, (*q1[u])[1], (AC)[5]);                                           \
            (AC)[6] = fmaf(f, (*q1[u])[2], (AC)[6]); (AC)[7] = fmaf(f, (*q1[u])[3], (AC)[7]);                                           \
            (AC)[8] = fmaf(f, (*q2[u])[0], (AC)[8]); (AC)[9] = fmaf(f, (*q2[u])[1], (AC)[9]);                                           \
        }                                                                                                                              \
    }

template <int MODE>
__global__ __launch_bounds__(512, 4) void tile_kernel(const float *__restrict__ f1, float *__restrict__ out, int tiles_per_wg, long long *cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *f0s = reinterpret_cast<float *>(smem + kStage);   // [64 cells][32 ch + 4]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(f1), 0, (int)((long)kImages * kC * kH * kW * 4), 0x00020000);
    for (int e = tid; e < 64 * 32; e += 512) f0s[(e >> 5) * 36 + (e & 31)] = f0_of(e >> 5, e & 31);
    long long t_fill = 0, t_d = 0;
    for (int it = 0; it < tiles_per_wg; ++it) {
        const int t = blockIdx.x * tiles_per_wg + it;
        const Tile tl = tile_of(t);
        // per-lane source offsets of the four DMA instructions of a plane (the same for every plane of the tile)
        unsigned voff[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = 64 * k + lane, row = i / kLQ, q = i - row * kLQ;
            voff[k] = (row < kRH && q < kNQ) ? (unsigned)(((tl.y0 + row) * kW + tl.x0 + 4 * q) * 4) : kOOB;   // past the region: the range check writes zeros
        }
        float acc[MODE == 1 ? 12 : (MODE >= 2 ? 20 : 1)];
#pragma unroll
        for (int a = 0; a < (int)(sizeof(acc) / sizeof(float)); ++a) acc[a] = 0.f;
        // MODE 1: 16 lanes per cell, 2 rounds of 32 cells; items s + 16 p, p = 0..2 -> (row = item >> 2, quad = item & 3)
        // MODE 2: 16 lanes per cell, lanes 0..9 = patch rows; MODE 3: item = tid + 512 p -> (cell = item / 10, row = item % 10)
        for (int ch = 0; ch < 2; ++ch) {
            const long long c0 = __builtin_readcyclecounter();
            __syncthreads();   // everyone is done reading the previous chunk
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const int p = wave * 2 + pp;   // plane of the chunk
                const unsigned so = (unsigned)(((long)tl.img * kC + ch * kChunk + p) * kH * kW * 4);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(smem + p * kPlaneBytes + k * 1024), 16, (int)voff[k], (int)so, 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0): the DMA of this wave has landed
            __syncthreads();
            const long long c1 = __builtin_readcyclecounter();
            t_fill += c1 - c0;
            if (MODE == 1) {
#pragma unroll
                for (int rd = 0; rd < 2; ++rd) {
                    const int cell = rd * 32 + wave * 4 + (lane >> 4), s = lane & 15;
                    int X, Y;
                    cell_origin(cell >> 4, cell & 15, X, Y);
                    const int xa = X & ~3;                      // aligned start of the 16-pixel window
                    const float *fq = f0s + cell * 36 + ch * kChunk;
#pragma unroll
                    for (int p3 = 0; p3 < 3; ++p3) {
                        const int item = s + 16 * p3, r = item >> 2, q = item & 3;
                        const bool on = item < 40;
                        const unsigned a = on ? (unsigned)(((Y + r) * kLP + xa + 4 * q) * 4) : 0u;
#pragma unroll
                        for (int c = 0; c < kChunk; ++c) {
                            const f32x4 v = *reinterpret_cast<const f32x4 *>(smem + c * kPlaneBytes + a);
                            const float f = fq[c];
                            float *ac = acc + (rd == 0 ? 0 : 0);   // (both rounds accumulate into the same registers: timing and a checksum)
                            ac[4 * p3 + 0] = fmaf(f, v[0], ac[4 * p3 + 0]);
                            ac[4 * p3 + 1] = fmaf(f, v[1], ac[4 * p3 + 1]);
                            ac[4 * p3 + 2] = fmaf(f, v[2], ac[4 * p3 + 2]);
                            ac[4 * p3 + 3] = fmaf(f, v[3], ac[4 * p3 + 3]);
                        }
                    }
                }
            } else if (MODE == 2) {
#pragma unroll
                for (int rd = 0; rd < 2; ++rd) {
                    const int cell = rd * 32 + wave * 4 + (lane >> 4), r = lane & 15;
                    int X, Y;
                    cell_origin(cell >> 4, cell & 15, X, Y);
                    const bool on = r < kPW;
                    const unsigned a = on ? (unsigned)(((Y + r) * kLP + X) * 4) : 0u;
                    const float *fq = f0s + cell * 36 + ch * kChunk;
#pragma unroll
                    for (int c = 0; c < kChunk; c += 4) ROW_BATCH(a, c, fq, acc + 10 * rd)
                }
            } else if (MODE == 3) {
#pragma unroll
                for (int p2 = 0; p2 < 2; ++p2) {
                    const int item = tid + 512 * p2;
                    if (item < 640) {   // (pass 1: the first two waves only)
                        const int cell = item / kPW, r = item - cell * kPW;
                        int X, Y;
                        cell_origin(cell >> 4, cell & 15, X, Y);
                        const unsigned a = (unsigned)(((Y + r) * kLP + X) * 4);
                        const float *fq = f0s + cell * 36 + ch * kChunk;
#pragma unroll
                        for (int c = 0; c < kChunk; c += 4) ROW_BATCH(a, c, fq, acc + 10 * p2)
                    }
                }
            }
            t_d += __builtin_readcyclecounter() - c1;
        }
        // results: MODE 2 / 3 write D[cell][row][0..9]; MODE 1 a per-lane checksum; MODE 0 a few staged values (the fill's layout)
        float *o = out + (size_t)t * 64 * 100;
        if (MODE == 0) {
            if (tid < 64) o[tid] = reinterpret_cast<float *>(smem)[(tid >> 2) * 1024 + ((tid & 3) == 3 ? 1000 : (tid & 3) * 209 + 3)];   // chunk 1, planes 0..15, scattered floats
        } else if (MODE == 1) {
            float sum = 0.f;
#pragma unroll
            for (int a = 0; a < 12; ++a) sum += acc[a];
            o[tid] = sum;
        } else if (MODE == 2) {
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
                const int cell = rd * 32 + wave * 4 + (lane >> 4), r = lane & 15;
                if (r < kPW)
#pragma unroll
                    for (int x = 0; x < kPW; ++x) o[cell * 100 + r * 10 + x] = acc[10 * rd + x];
            }
        } else {
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
                const int item = tid + 512 * p2;
                if (item < 640)
#pragma unroll
                    for (int x = 0; x < kPW; ++x) o[item * 10 + x] = acc[10 * p2 + x];
            }
        }
    }
    if (tid == 0) { cyc[2 * blockIdx.x] = t_fill; cyc[2 * blockIdx.x + 1] = t_d; }
}

template <int MODE>
void run(const float *d_f1, const std::vector<float> &h_f1, float *d_out, long long *d_cyc, int wgs, int tpw) {
    const size_t lds = kStage + 64 * 36 * 4;
    hipFuncSetAttribute(reinterpret_cast<const void *>(tile_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(tile_kernel<MODE>, dim3(wgs), dim3(512), lds, 0, d_f1, d_out, tpw, d_cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 10;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(tile_kernel<MODE>, dim3(wgs), dim3(512), lds, 0, d_f1, d_out, tpw, d_cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> cyc(2 * wgs);
    hipMemcpy(cyc.data(), d_cyc, sizeof(long long) * 2 * wgs, hipMemcpyDeviceToHost);
    double cf = 0, cd = 0;
    for (int b = 0; b < wgs; ++b) { cf += cyc[2 * b]; cd += cyc[2 * b + 1]; }
    cf /= (double)wgs * tpw; cd /= (double)wgs * tpw;
    // check the tiles of workgroup 0
    std::vector<float> o((size_t)tpw * 6400);
    hipMemcpy(o.data(), d_out, o.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    auto px = [&](int img, int c, int y, int x) { return h_f1[(((size_t)img * kC + c) * kH + y) * kW + x]; };
    for (int t = 0; t < tpw && MODE >= 2; ++t) {
        const Tile tl = tile_of(t);
        for (int cell = 0; cell < 64; ++cell) {
            int X, Y;
            cell_origin(cell >> 4, cell & 15, X, Y);
            for (int r = 0; r < 10; ++r)
                for (int x = 0; x < 10; ++x) {
                    float a = 0.f;
                    for (int c = 0; c < kC; ++c) a = fmaf(f0_of(cell, c), px(tl.img, c, tl.y0 + Y + r, tl.x0 + X + x), a);
                    if (o[(size_t)t * 6400 + cell * 100 + r * 10 + x] != a) ++bad;
                }
        }
    }
    if (MODE == 0) {   // the fill's layout: plane p (chunk 1 -> channel 16 + p), float index 209 j + 3 of the plane
        const Tile tl = tile_of(tpw - 1);
        for (int e = 0; e < 64; ++e) {
            const int p = e >> 2, fi = (e & 3) == 3 ? 1000 : (e & 3) * 209 + 3, row = fi / kLP, x = fi % kLP;
            const float want = (row < kRH && x < kRW) ? px(tl.img, 16 + p, tl.y0 + row, tl.x0 + x) : 0.f;   // floats 836..1023 of a plane: zeros from the range check
            if (o[(size_t)(tpw - 1) * 6400 + e] != want) ++bad;
        }
    }
    printf("MODE %d: %8.1f us per launch of %d tiles (%.2f us per tile and CU-slot);  cycles per tile: fill (2 chunks, incl. barriers) %7.0f, D-stage %7.0f;  "
           "check: %s\n", MODE, ms * 1e3 / reps, wgs * tpw, ms * 1e3 / reps / tpw, cf, cd, MODE == 1 ? "(timing only)" : (bad ? "MISMATCH" : "ok"));
    if (bad) printf("   %d mismatches\n", bad);
}

int main(int argc, char **argv) {
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    const size_t n = (size_t)kImages * kC * kH * kW;
    std::vector<float> h(n);
    unsigned s = 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 9) - (1 << 22)) / (float)(1 << 21); }
    float *d_f1, *d_out;
    long long *d_cyc;
    const int wgs = 512, tpw = 8;   // 4 096 tiles: the roofline op's count
    hipMalloc(&d_f1, n * 4);
    hipMalloc(&d_out, (size_t)wgs * tpw * 6400 * 4);
    hipMalloc(&d_cyc, sizeof(long long) * 2 * wgs);
    hipMemcpy(d_f1, h.data(), n * 4, hipMemcpyHostToDevice);
    if (only < 0 || only == 0) run<0>(d_f1, h, d_out, d_cyc, wgs, tpw);
    if (only < 0 || only == 1) run<1>(d_f1, h, d_out, d_cyc, wgs, tpw);
    if (only < 0 || only == 2) run<2>(d_f1, h, d_out, d_cyc, wgs, tpw);
    if (only < 0 || only == 3) run<3>(d_f1, h, d_out, d_cyc, wgs, tpw);
    return 0;
}
