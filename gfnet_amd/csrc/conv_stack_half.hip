// conv_stack_half.hip -- the refiners' conv blocks on fp16 maps (the reference's amp=True class).
//
// Reference: ConvRefiner.forward, model/network.py:560-562 -- `with torch.autocast("cuda", enabled=self.amp, dtype=self.amp_dtype)`
// around block1 + hidden_blocks: every map between two blocks is a float16 tensor there.  Same fused kernel as
// csrc/conv_stack.hip's fp16-operand variant (csrc/conv_block_fused.h: depthwise, BatchNorm and accumulation in fp32, 1x1
// operands fp16) with the maps between blocks stored as fp16 in HBM, channel pairs side by side ((B, ceil(C/2), G, G) half2):
// the blocks of the fine scales (C = 24, 73 on 128^2 .. 320^2 maps) are bound by that traffic, and it halves.
// The first block of a stack reads the fp32 concat `d`, the last one (out_conv folded in) writes fp32.
#include "conv_block_fused.h"

namespace {

// GFN_CONV_VALU_DW (environment, experiments): the depthwise on the VALU (fp32 taps) instead of the matrix core
static const bool g_valu_dw = gfn::exp_env("GFN_CONV_VALU_DW") != nullptr;
static const int g_tw16_min = gfn::exp_env("GFN_CONV_TW16_MIN") ? atoi(gfn::exp_env("GFN_CONV_TW16_MIN")) : 0;

// GFN_CONV_KW1: one K tile per iteration for the wide blocks too (experiments)
static const bool g_kw1 = gfn::exp_env("GFN_CONV_KW1") != nullptr;

template <bool HIN, bool HOUT, bool MM>
int launch_half(const void *x, const float *packed, void *y, int B, int C, int M, int G, int dbg, hipStream_t s) {
    const float *xf = (const float *)x;
    float *yf = (float *)y;
    if constexpr (HIN && HOUT && MM) {  // blocks between two half maps, wide enough for more than three accumulator row tiles
        if (M > 96 && !g_kw1) {
            if (G % 32 == 0 || G > 160) return launch_fused<32, true, HIN, HOUT, MM, 2>(xf, packed, yf, B, M, C, G, dbg, s);
            if (G % 16 == 0 || G > (g_tw16_min ? g_tw16_min : 64)) return launch_fused<16, true, HIN, HOUT, MM, 2>(xf, packed, yf, B, M, C, G, dbg, s);
            return launch_fused<8, true, HIN, HOUT, MM, 2>(xf, packed, yf, B, M, C, G, dbg, s);
        }
    }
    if (G % 32 == 0 || G > 160) return launch_fused<32, true, HIN, HOUT, MM>(xf, packed, yf, B, M, C, G, dbg, s);
    if (G % 16 == 0 || G > (g_tw16_min ? g_tw16_min : 64)) return launch_fused<16, true, HIN, HOUT, MM>(xf, packed, yf, B, M, C, G, dbg, s);
    return launch_fused<8, true, HIN, HOUT, MM>(xf, packed, yf, B, M, C, G, dbg, s);
}

}  // namespace

GFN_EXPORT int gfn_conv_block_half_fwd(const void *x, int x_dtype, const float *packed, void *y, int y_dtype, int B, int C, int M,
                                       int G, gfn_stream_t stream) {
    if (!x || !packed || !y || B < 0 || C <= 0 || M <= 0 || G <= 0) return gfn::fail(GFN_ERR_INVALID_ARG, "conv_block_half: bad argument");
    int dbg = 0;
#ifdef GFN_ABLATE  // timing experiments (tools/ablate_convblock.py): phase mask in the high bits of x_dtype
    dbg = x_dtype >> 8;
    x_dtype &= 0xff;
#endif
    if ((x_dtype != GFN_F32 && x_dtype != GFN_F16) || (y_dtype != GFN_F32 && y_dtype != GFN_F16))
        return gfn::fail(GFN_ERR_INVALID_ARG, "conv_block_half: map dtypes must be GFN_F32 or GFN_F16");
    if (x_dtype == GFN_F32 && y_dtype == GFN_F32)
        return gfn::fail(GFN_ERR_INVALID_ARG, "conv_block_half: fp32 in and out is gfn_conv_block_fwd (variant 2)");
    if (x == y) return gfn::fail(GFN_ERR_INVALID_ARG, "conv_block_half: in-place is not supported (cells read their neighbours)");
    if (G & 3) return gfn::fail(GFN_ERR_INVALID_ARG, "conv_block_half: grid side must be a multiple of 4 (got %d)", G);
    if ((long)C * G * G > 0x1fffffffL) return gfn::fail(GFN_ERR_INVALID_ARG, "conv_block_half: a map (C*G*G elements) must stay below 2 GB");
    if (B == 0) return GFN_OK;
    hipStream_t s = (hipStream_t)stream;
    if (g_valu_dw) {
        if (x_dtype == GFN_F32) return launch_half<false, true, false>(x, packed, y, B, C, M, G, dbg, s);
        if (y_dtype == GFN_F32) return launch_half<true, false, false>(x, packed, y, B, C, M, G, dbg, s);
        return launch_half<true, true, false>(x, packed, y, B, C, M, G, dbg, s);
    }
    if (x_dtype == GFN_F32) return launch_half<false, true, true>(x, packed, y, B, C, M, G, dbg, s);
    if (y_dtype == GFN_F32) return launch_half<true, false, true>(x, packed, y, B, C, M, G, dbg, s);
    return launch_half<true, true, true>(x, packed, y, B, C, M, G, dbg, s);
}
