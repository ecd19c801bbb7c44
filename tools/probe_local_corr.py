"""Timing probe (GPU): local-correlation kernel on the production shapes. Not part of the product."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import synth  # noqa: E402
from gfnet_amd.utils.local_correlation import local_correlation  # noqa: E402

SHAPES = [(64, 32, 32, 7), (64, 56, 32, 6), (32, 112, 64, 4), (16, 224, 128, 2),
          (64, 70, 40, 6), (32, 140, 80, 4), (16, 280, 160, 2)]


def algo_bytes(B, c, hs, G, r):
    K = (2 * r + 1) ** 2
    return 4 * B * (c * G * G + c * hs * hs + 2 * G * G + K * G * G)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    kinds = sys.argv[2].split(",") if len(sys.argv) > 2 else ["homography"]
    variant = int(os.environ.get("VARIANT", "0"))
    for kind in kinds:
        for (c, hs, G, r) in SHAPES:
            f0 = torch.randn(B, c, G, G, device="cuda")
            f1 = torch.randn(B, c, hs, hs, device="cuda")
            if kind == "homography":
                flow = torch.from_numpy(np.tile(synth.homography_flow(2, G, 5), (B // 2, 1, 1, 1))).cuda()
            elif kind == "identity":
                lin = torch.linspace(-1 + 1 / G, 1 - 1 / G, G, device="cuda")
                gy, gx = torch.meshgrid(lin, lin, indexing="ij")
                flow = torch.stack((gx, gy))[None].repeat(B, 1, 1, 1).contiguous()
            else:
                flow = torch.rand(B, 2, G, G, device="cuda") * 1.8 - 0.9
            out = torch.empty(B, (2 * r + 1) ** 2, G, G, device="cuda")
            for _ in range(3):
                local_correlation((B, c, hs, hs), f0, f1, r, G, flow=flow, out=out, _variant=variant)
            torch.cuda.synchronize()
            n = 20
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                local_correlation((B, c, hs, hs), f0, f1, r, G, flow=flow, out=out, _variant=variant)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
            by = algo_bytes(B, c, hs, G, r)
            print(f"{kind:10s} c{c:<3d} hs{hs:<4d} G{G:<4d} r{r}  B={B}: {us:9.1f} us  {by/us/1e3:8.1f} GB/s algorithmic "
                  f"({by/us/1e3/8000*100:5.1f}% of 8 TB/s)", flush=True)


if __name__ == "__main__":
    main()
