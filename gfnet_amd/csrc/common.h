// Shared host-side helpers for the C-ABI entry points (gfx950 only; no CUDA dual path).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "../../include/gfnet_hip.h"

#define GFN_EXPORT extern "C" __attribute__((visibility("default")))

namespace gfn {

char *last_error_buf();  // thread-local, 512 bytes
int fail(int code, const char *fmt, ...);

// Experiment switches (GFN_CONV_*, GFN_KDE_* environment variables read by tools/): only the -DGFN_ABLATE build
// (python -m gfnet_amd.build --ablate -> libgfnet_hip_ablate.so) looks at the environment; in the product library the environment
// cannot change which kernel runs.
inline const char *exp_env(const char *name) {
#ifdef GFN_ABLATE
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GFN_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return GFN_OK;
}

// Bijective XCD-aware remap of a 1-D block id (guide T1): blocks are dealt round-robin over the
// 8 XCDs, so block ids congruent mod 8 share an L2.  Give every XCD one contiguous run of the
// work list, so that consecutive work items (tiles of the same image) hit the same L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned n) {
    const unsigned q = n >> 3, rem = n & 7u;
    const unsigned xcd = bid & 7u, local = bid >> 3;
    const unsigned start = xcd * q + (xcd < rem ? xcd : rem);
    return start + local;
}

// The same with the step (end - start) / (steps - 1) supplied by the caller (computed once on the host in fp32: an IEEE
// division there and here give the same bits; on the device it costs a dozen vector instructions per call).
__device__ __forceinline__ float linspace_step_at(float start, float end, float step, int steps, int i) {
    if (steps == 1) return start;
    return (i < steps / 2) ? start + step * (float)i : end - step * (float)(steps - i - 1);
}

// torch.linspace(start, end, steps)[i] in fp32, the way ATen fills it (from both ends).
__device__ __forceinline__ float linspace_at(float start, float end, int steps, int i) {
    if (steps == 1) return start;
    const float step = (end - start) / (float)(steps - 1);
    return (i < steps / 2) ? start + step * (float)i : end - step * (float)(steps - i - 1);
}

}  // namespace gfn
