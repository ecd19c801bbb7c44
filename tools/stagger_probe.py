"""Experiment (GFN_ABLATE build): stagger the first dispatch wave of the tiled local-correlation kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["GFNET_HIP_LIB"] = os.path.join(ROOT, "gfnet_amd", "csrc", "libgfnet_hip_ablate.so")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch, synth
from gfnet_amd.utils.local_correlation import local_correlation
B, c, hs, G, r = 64, 32, 112, 64, 4
f0 = torch.randn(B, c, G, G, device="cuda"); f1 = torch.randn(B, c, hs, hs, device="cuda")
flow = torch.from_numpy(np.tile(synth.homography_flow(2, G, 5), (B // 2, 1, 1, 1))).cuda()
out = torch.empty(B, (2 * r + 1) ** 2, G, G, device="cuda")
for mode in (0, 64, 128, 256):
    for n in ((0,) if mode == 0 else (1, 2, 3, 4)):
        v = (mode | (n << 12)) << 8
        for _ in range(12):
            local_correlation((B, c, hs, hs), f0, f1, r, G, flow=flow, out=out, _variant=v)
        torch.cuda.synchronize()
        print("mode", mode, "sleep", n, flush=True)
