"""refiner_input alone (symmetric batch of 32 pairs = 64 directions), for PMC runs: python tools/probe_refiner_input.py C HS G DD [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gfnet_amd import ops  # noqa: E402

c, hs, G, dd = (int(v) for v in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
a = torch.randn(32, c, hs, hs, device="cuda")
b = torch.randn(32, c, hs, hs, device="cuda")
lin = torch.linspace(-1 + 1 / G, 1 - 1 / G, G, device="cuda")
gy, gx = torch.meshgrid(lin, lin, indexing="ij")
flow = (torch.stack((gx, gy))[None] * 0.9).repeat(64, 1, 1, 1).contiguous()
wgt = torch.randn(dd, 2, 1, 1, device="cuda")
bias = torch.randn(dd, device="cuda")
for _ in range(reps):
    ops.refiner_input(G, a, b, flow, wgt, bias, 0, corr_in_other=False)
torch.cuda.synchronize()
print("done")
