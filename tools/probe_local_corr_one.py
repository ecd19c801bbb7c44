"""One local-correlation shape, a few calls (for PMC runs): python tools/probe_local_corr_one.py C HS G R [B] [reps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import synth  # noqa: E402
from gfnet_amd.utils.local_correlation import local_correlation  # noqa: E402

c, hs, G, r = (int(v) for v in sys.argv[1:5])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 64
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 4
f0 = torch.randn(B, c, G, G, device="cuda")
f1 = torch.randn(B, c, hs, hs, device="cuda")
flow = torch.from_numpy(np.tile(synth.homography_flow(2, G, 5), (B // 2, 1, 1, 1))).cuda()
out = torch.empty(B, (2 * r + 1) ** 2, G, G, device="cuda")
variant = int(os.environ.get("VARIANT", "0"))  # 0: lean tile kernel (r <= 4), 2: round-1 tile kernel
for _ in range(reps):
    local_correlation((B, c, hs, hs), f0, f1, r, G, flow=flow, out=out, _variant=variant)
torch.cuda.synchronize()
print("done")
