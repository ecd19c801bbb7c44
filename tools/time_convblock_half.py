"""Timing (GPU): one half-map conv block per refiner width; GFN_CONV_TPB / GFN_CONV_NB1 in the environment select launch shapes
(read only by the -DGFN_ABLATE build, python -m gfnet_amd.build --ablate, which this tool loads when it exists)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_abl = os.path.join(ROOT, "gfnet_amd", "csrc", "libgfnet_hip_ablate.so")
_prod = os.path.join(ROOT, "gfnet_amd", "csrc", "libgfnet_hip.so")
if os.path.exists(_abl) and "GFNET_HIP_LIB" not in os.environ:
    if os.path.exists(_prod) and os.path.getmtime(_abl) < os.path.getmtime(_prod):  # never time a stale experiment build silently
        sys.exit(f"{_abl} is older than {_prod}: rebuild it (python -m gfnet_amd.build --ablate) or set GFNET_HIP_LIB")
    os.environ["GFNET_HIP_LIB"] = _abl
print("library:", os.environ.get("GFNET_HIP_LIB", _prod), flush=True)
import torch
sys.path.insert(0, ROOT)
from gfnet_amd import ops
B = 64
shapes = [(417, 32), (417, 40), (361, 32), (361, 40), (177, 64), (177, 80), (73, 128), (73, 160), (24, 256), (24, 320)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
out = []
for C, G in shapes:
    packed = ops.conv_block_pack(torch.randn(C, 25, device="cuda") * 0.2, torch.randn(C, device="cuda"), torch.rand(C, device="cuda") + 0.5,
                                 torch.randn(C, device="cuda"), torch.randn(C, C, device="cuda") * C ** -0.5, torch.randn(C, device="cuda"))
    xh = torch.randn(B, (C + 1) // 2, G, G, 2, device="cuda").half()
    yh = torch.empty_like(xh)
    best = 1e9
    for rnd in range(3):
        for _ in range(2):
            ops.conv_block_half(xh, packed, C, C, out=yh)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.conv_block_half(xh, packed, C, C, out=yh)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5 * 1e3)
    mb = 2 * B * ((C + 1) // 2) * G * G * 4 / 1e6
    out.append(f"c{C}g{G} {best:.0f}us {mb / best:.2f}TB/s")
print("tpb=%s nb1=%s: " % (os.environ.get("GFN_CONV_TPB", "-"), os.environ.get("GFN_CONV_NB1", "-")) + " | ".join(out), flush=True)
