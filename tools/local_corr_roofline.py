"""Per-launch table of every local-correlation shape of a bench step: calls per step, mean microseconds of the C-ABI call
(HIP events on its stream, as bench.py's roofline field), algorithmic bytes (f0 + f1 + flow + out), fraction of
the 8 TB/s HBM peak.  usage (GPU box): python tools/local_corr_roofline.py [--workload 448b32|672b16|pyr-fp16] > table.md"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gfnet_amd import _synthetic as bench  # noqa: E402
from gfnet_amd import ops  # noqa: E402

wl_key = sys.argv[sys.argv.index("--workload") + 1] if "--workload" in sys.argv else "448b32"
wl = bench.WORKLOADS[wl_key]
dev = torch.device("cuda:0")
dtype = torch.float16 if wl["dtype"] == "fp16" else torch.float32
scenes = [bench.Scene(S, wl["pairs"], wl["num_itr"], dtype, "off", dev, 0) for S in wl["sizes"]]


class _All(dict):  # time every local_corr_* call
    def __contains__(self, k):
        return k.startswith("local_corr_")

    def __missing__(self, k):
        self[k] = []
        return self[k]


def match(sc):
    return sc.model.match_pyramids(sc.pyr[0], sc.pyr[1], sc.pyr_up[0], sc.pyr_up[1], batched=True)


for _ in range(3):
    for sc in scenes:
        match(sc)
torch.cuda.synchronize()
ops.kernel_events = _All()
steps = 10
for _ in range(steps):
    for sc in scenes:
        match(sc)
torch.cuda.synchronize()
ev, ops.kernel_events = ops.kernel_events, None
fb = 2 if dtype == torch.float16 else 4
print(f"# local correlation, per C-ABI call: workload {wl_key} ({wl['label']})\n")
print("| shape (c, f1 side, grid, r) | calls/step | mean µs | algorithmic MB | GB/s | frac of 8 TB/s |")
print("|---|---|---|---|---|---|")
B2 = 2 * wl["pairs"]
for name in sorted(ev, key=lambda n: -sum(a.elapsed_time(b) for a, b in ev[n])):
    f = dict((p[0], int(p[1:])) for p in name.split("_")[2:])
    us = sum(a.elapsed_time(b) for a, b in ev[name]) / len(ev[name]) * 1e3
    K = (2 * f["r"] + 1) ** 2
    nbytes = B2 * (f["c"] * f["g"] ** 2 * 4 + f["c"] * f["h"] ** 2 * fb + 8 * f["g"] ** 2 + K * f["g"] ** 2 * 4)
    print(f"| c{f['c']}, {f['h']}², G{f['g']}, r{f['r']} | {len(ev[name]) / steps:g} | {us:.1f} | {nbytes / 1e6:.1f} | {nbytes / us / 1e3:.0f} | "
          f"{nbytes / us / 1e3 / 8000:.3f} |")
print("\nf0 is the fp32 grid_feature slice of the concat buffer, f1 the feature map as stored (fp32 or fp16), out fp32; "
      f"{B2} directions per call; flows = the bench's noisy homography flows.")
