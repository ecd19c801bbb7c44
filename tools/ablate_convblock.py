"""Timing experiment (GPU, ablation build): the fused conv block with parts switched off.
Build first:  python -m gfnet_amd.build --ablate ;  run with GFNET_HIP_LIB=gfnet_amd/csrc/libgfnet_hip_ablate.so"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gfnet_amd import ops

MASKS = [(0, "full"), (1, "-global loads"), (2, "-depthwise"), (4, "-mfma"), (8, "-stores"), (16, "-commit"), (32, "-zero"), (2 | 4, "-dw -mfma"),
         (1 | 16, "-loads -commit"), (1 | 2 | 16, "only mfma+stores"), (1 | 4 | 16 | 8, "only depthwise"), (1 | 2 | 4 | 16, "only stores"),
         (2 | 4 | 8, "only loads+commit"), (63, "nothing")]
PREC = int(os.environ.get("ABLATE_VARIANT", "2"))
HALF = os.environ.get("ABLATE_HALF", "0") == "1"  # fp16 maps (gfn_conv_block_half_fwd, half in and out)
from gfnet_amd import _lib
from gfnet_amd._lib import ptr, stream_ptr
shapes = [(417, 32), (177, 64), (73, 128), (24, 256)]
B = 64
for C, G in shapes:
    x = torch.randn(B, C, G, G, device="cuda")
    packed = ops.conv_block_pack(torch.randn(C, 25, device="cuda") * 0.2, torch.randn(C, device="cuda"), torch.rand(C, device="cuda") + 0.5,
                                 torch.randn(C, device="cuda"), torch.randn(C, C, device="cuda") * C ** -0.5, torch.randn(C, device="cuda"))
    y = torch.empty_like(x)
    res = {}
    if HALF:
        xh = torch.randn(B, (C + 1) // 2, G, G, 2, device="cuda").half()
        yh = torch.empty_like(xh)

        def run(m):
            _lib.lib().gfn_conv_block_half_fwd(ptr(xh), (m << 8) | 1, ptr(packed), ptr(yh), 1, B, C, C, G, stream_ptr(xh.device))
    else:
        def run(m):
            ops.conv_block(x, packed, C, out=y, variant=(m << 8) | PREC)
    for rnd in range(3):
        for m, name in MASKS:
            for _ in range(2):
                run(m)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run(m)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) / 5 * 1e3)
    print(f"{'half maps' if HALF else 'variant %d' % PREC} C={C} G={G}: " + " | ".join(f"{n} {min(v):.0f}" for n, v in res.items()), flush=True)
