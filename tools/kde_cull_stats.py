"""How many (64-query wave, 64-point block) pairs survive the bounding-box cull of kde4_mfma_kernel on the bench's own
sample stage?  Captures the KDE input of one bench step and redoes the cull test in torch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gfnet_amd import ops  # noqa: E402

captured = []
orig = ops.kde_density


def spy(x, *a, **k):
    captured.append(x.detach().clone())
    return orig(x, *a, **k)


ops.kde_density = spy
sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--cpu-pairs", "0"]
bench.main()
x = captured[0]
Bt, N, _ = x.shape
xs, _ = ops._morton_sorted(x, x.device)
scale2 = 1.4426950408889634 / (2 * 0.1 * 0.1)
nb = (N + 63) // 64
pad = nb * 64 - N
if pad:
    xs = torch.cat([xs, xs[:, -1:].expand(Bt, pad, 4)], 1)
blk = xs.view(Bt, nb, 64, 4)
lo, hi = blk.amin(2), blk.amax(2)  # (Bt, nb, 4)
gap = torch.clamp(torch.maximum(lo[:, :, None] - hi[:, None, :], lo[:, None, :] - hi[:, :, None]), min=0)
d2 = (gap * gap).sum(-1) * scale2
for cut in (24.0, 32.0, 40.0):
    print(f"cutoff 2^-{cut:.0f}: {float((d2 <= cut).float().mean()):.3f} of block pairs survive (N={N}, Bt={Bt}, {nb} blocks/row)")
ext = (hi - lo)
print("mean block extent per dim:", ext.mean((0, 1)).tolist())
# ideal: point pairs actually within the cutoff
sub = xs[0, :4096]
dd = torch.cdist(sub, sub) ** 2 * scale2
print(f"point pairs within 2^-32: {float((dd <= 32).float().mean()):.3f} (first 4096 points of row 0: local subset, upper bound)")
idx = torch.randperm(N, device=x.device)[:4096]
sub = xs[0, idx]
dd = torch.cdist(sub, sub) ** 2 * scale2
print(f"point pairs within 2^-32: {float((dd <= 32).float().mean()):.3f} (random 4096 points of row 0)")


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"kde_density on the captured input: {timeit(lambda: orig(x, std=0.1)):.1f} us (sort + operands + kernel + scatter)")
