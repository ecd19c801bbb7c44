"""Wall time of the refiner-input launches of a 448b32 step (GPU): python tools/time_refiner_input.py
Shapes: scale 1 of both passes (no plan) and the coarser scales without their plan blocks; 64 directions, flows = 0.9 x identity."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gfnet_amd import ops  # noqa: E402

torch.manual_seed(0)
for (c, hs, G, dd) in [(8, 448, 256, 8), (8, 560, 320, 8), (16, 224, 128, 16), (16, 280, 160, 16), (32, 112, 64, 32), (64, 56, 32, 64)]:
    a = torch.randn(32, c, hs, hs, device="cuda")
    b = torch.randn(32, c, hs, hs, device="cuda")
    lin = torch.linspace(-1 + 1 / G, 1 - 1 / G, G, device="cuda")
    gy, gx = torch.meshgrid(lin, lin, indexing="ij")
    flow = (torch.stack((gx, gy))[None] * 0.9).repeat(64, 1, 1, 1).contiguous()
    wgt = torch.randn(dd, 2, 1, 1, device="cuda")
    bias = torch.randn(dd, device="cuda")
    for _ in range(3):
        d = ops.refiner_input(G, a, b, flow, wgt, bias, 0, corr_in_other=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        d = ops.refiner_input(G, a, b, flow, wgt, bias, 0, corr_in_other=False)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    nbytes = 64 * (c * hs * hs * 4) + 64 * (2 * c + dd) * G * G * 4
    print(f"c{c} {hs}^2 G{G}: {us:8.1f} us   unique bytes {nbytes / 1e6:7.1f} MB -> {nbytes / us / 1e6:5.2f} TB/s   checksum {float(d.double().sum()):.6f}", flush=True)
