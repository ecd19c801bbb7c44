"""Stage-ablation timing of the lean local-correlation tile kernel (r <= 4; needs the GFN_ABLATE build: python -m gfnet_amd.build --ablate).
Results with a stage switched off are wrong by construction; only the times mean anything."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["GFNET_HIP_LIB"] = os.path.join(ROOT, "gfnet_amd", "csrc", "libgfnet_hip_ablate.so")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import synth  # noqa: E402
from gfnet_amd.utils.local_correlation import local_correlation  # noqa: E402

NAMES = ((1, "stage"), (2, "dstage"), (4, "f0"), (8, "epilogue"), (16, "dbuf"), (32, "table"), (64, "rows7"), (128, "stores"))
SHAPES = [(32, 112, 64, 4), (16, 224, 128, 2)]
MASKS = [0, 1, 2, 4, 8, 128, 16, 32, 64, 1 | 2, 2 | 8, 1 | 8, 1 | 2 | 8, 64 | 2, 64 | 8, 1 | 2 | 4 | 8 | 16 | 32]
B = 64
for (c, hs, G, r) in SHAPES:
    f0 = torch.randn(B, c, G, G, device="cuda")
    f1 = torch.randn(B, c, hs, hs, device="cuda")
    flow = torch.from_numpy(np.tile(synth.homography_flow(2, G, 5), (B // 2, 1, 1, 1))).cuda()
    out = torch.empty(B, (2 * r + 1) ** 2, G, G, device="cuda")
    for m in MASKS:
        v = m << 8
        for _ in range(3):
            local_correlation((B, c, hs, hs), f0, f1, r, G, flow=flow, out=out, _variant=v)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            local_correlation((B, c, hs, hs), f0, f1, r, G, flow=flow, out=out, _variant=v)
        e1.record()
        torch.cuda.synchronize()
        names = [n for b, n in NAMES if m & b]
        print(f"c{c} hs{hs} G{G} r{r}: skip[{'+'.join(names) or 'nothing':40s}] {e0.elapsed_time(e1)*50:8.1f} us", flush=True)
