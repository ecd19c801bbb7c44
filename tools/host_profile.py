"""Where the HOST time of a step goes (GPU box): python tools/host_profile.py [workload]  -- cProfile over eager steps of a workload's scenes
on one stream; the three-scene workload (pyr-fp16) is bound by the host's launch rate."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gfnet_amd import _synthetic as bench  # noqa: E402

wl_key = sys.argv[1] if len(sys.argv) > 1 else "pyr-fp16"
wl = bench.WORKLOADS[wl_key]
dev = torch.device("cuda:0")
dtype = torch.float16 if wl["dtype"] == "fp16" else torch.float32
scenes = [bench.Scene(S, wl["pairs"], wl["num_itr"], dtype, "off", dev, 0) for S in wl["sizes"]]
for i in range(3):
    for sc in scenes:
        sc.step(i)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
n = 10
for i in range(n):
    for sc in scenes:
        sc.step(i)
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host time per step (all scenes): {(t1 - t0) / n * 1e3:.3f} ms (under the profiler); device caught up {(t2 - t1) * 1e3:.2f} ms later")
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
