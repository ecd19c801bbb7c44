"""Timing probe (GPU): homography solver stages, KDE, refiner_input, corr_softargmax."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gfnet_amd import ops  # noqa: E402
from test_homography_cpu import make_points, random_h  # noqa: E402


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


rng = np.random.default_rng(0)
Bt = 32
pts = torch.from_numpy(np.stack([make_points(rng, random_h(rng), 5000, noise=0.5, outliers=0.3) for _ in range(Bt)])).cuda()
for stage in (1, 2, 0):
    print(f"find_homography stage={stage}: {timeit(lambda: ops.find_homography(pts, iters=2000, stage=stage)):.3f} ms")
w = torch.rand(Bt, 5000, device="cuda")
print(f"homography_dlt: {timeit(lambda: ops.homography_dlt(pts, w)):.3f} ms")
x = torch.rand(Bt, 20000, 4, device="cuda") * 2 - 1
print(f"kde 32x20000^2: {timeit(lambda: ops.kde_density(x, std=0.1)):.3f} ms")
f0 = torch.randn(64, 64, 32, 32, device="cuda")
f1 = torch.randn(64, 64, 32, 32, device="cuda")
print(f"corr_softargmax 64x(64,32^2): {timeit(lambda: ops.corr_softargmax(f0, f1)):.3f} ms")
for (c, hs, G, dd, r) in [(64, 32, 32, 64, 7), (32, 112, 64, 32, 4), (16, 224, 128, 16, 2), (8, 448, 256, 8, 0), (8, 560, 320, 8, 0)]:
    a = torch.randn(64, c, hs, hs, device="cuda")
    b = torch.randn(64, c, hs, hs, device="cuda")
    lin = torch.linspace(-1 + 1 / G, 1 - 1 / G, G, device="cuda")
    gy, gx = torch.meshgrid(lin, lin, indexing="ij")
    flow = (torch.stack((gx, gy))[None] * 0.9).repeat(64, 1, 1, 1).contiguous()
    wgt = torch.randn(dd, 2, 1, 1, device="cuda")
    bias = torch.randn(dd, device="cuda")
    t = timeit(lambda: ops.refiner_input(G, a[:32], b[:32], flow, wgt, bias, r, corr_in_other=False))
    by = 4 * 64 * ((2 * c + dd) * G * G + 2 * c * hs * hs)
    print(f"refiner_input (no corr) c{c} hs{hs} G{G}: {t*1e3:.1f} us  ({by/t/1e6:.0f} GB/s of in+out bytes)")
