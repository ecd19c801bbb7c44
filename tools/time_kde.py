"""Time of the sampling stage's KDE on the bench's own matches (GPU): python tools/time_kde.py  -- captures the KDE input of one bench step and
times ops.kde_density (sort + operands + matrix-core kernel + combine) on it."""
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
    captured.append((x.detach().clone(), a, k))
    return orig(x, *a, **k)


ops.kde_density = spy
sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--cpu-pairs", "0", "--no-stack-leg", "--no-other-workloads", "--no-stress-legs"]
bench.main()
ops.kde_density = orig
x, a, k = captured[0]
for _ in range(3):
    d = orig(x, *a, **k)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 20
for _ in range(n):
    d = orig(x, *a, **k)
e1.record()
torch.cuda.synchronize()
print(f"kde_density on {tuple(x.shape)}: {e0.elapsed_time(e1) * 1e3 / n:.1f} us per call, checksum {float(d.double().sum()):.6f}")
