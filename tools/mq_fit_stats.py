"""Which 2 x 16-cell tiles of the bench's r >= 5 local-correlation calls fit the matrix-core tile kernel (csrc/local_corr_mq.h)
and why the others do not: python tools/mq_fit_stats.py [--workload 448b32]  (GPU box: the flows are the bench scenes' own)"""
import os
import sys

import numpy as np
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
seen = []
orig = ops.refiner_input


def spy(num_grid, x, y, flow, disp_w, disp_b, local_radius, **kw):
    if local_radius >= 5:
        seen.append((int(num_grid), x.shape[-2], x.shape[-1], int(local_radius), flow.detach().float().cpu().numpy()))
    return orig(num_grid, x, y, flow, disp_w, disp_b, local_radius, **kw)


ops.refiner_input = spy
import gfnet_amd.model.network as net  # noqa: E402
if hasattr(net, "ops"):
    net.ops.refiner_input = spy
for sc in scenes:
    sc.model.match_pyramids(sc.pyr[0], sc.pyr[1], sc.pyr_up[0], sc.pyr_up[1], batched=True)
torch.cuda.synchronize()
for G, H, W, r, fl in seen:
    PW = 2 * r + 2
    B = fl.shape[0]
    f32 = np.float32
    x0 = np.floor(((fl[:, 0] + f32(-2.0 * r / W) + f32(1)) * f32(W) - f32(1)) / f32(2)).astype(np.int64)
    y0 = np.floor(((fl[:, 1] + f32(-2.0 * r / H) + f32(1)) * f32(H) - f32(1)) / f32(2)).astype(np.int64)
    touch = (x0 < W) & (x0 + PW > 0) & (y0 < H) & (y0 + PW > 0)
    ty, tx = (G + 1) // 2, (G + 15) // 16
    pad = lambda a, fill: np.pad(a, ((0, 0), (0, ty * 2 - G), (0, tx * 16 - G)), constant_values=fill)
    big = 1 << 20
    lo_x, lo_y = pad(np.where(touch, x0, big), big), pad(np.where(touch, y0, big), big)
    hi_x, hi_y = pad(np.where(touch, x0 + PW, -big), -big), pad(np.where(touch, y0 + PW, -big), -big)
    def red(a, f, cols):
        return f(f(a.reshape(B, ty, 2, tx * 16 // cols, cols), axis=4), axis=2)
    gw = red(hi_x, np.max, 8) - red(lo_x, np.min, 8)
    gh = red(hi_y, np.max, 8) - red(lo_y, np.min, 8)
    tw = red(hi_x, np.max, 16) - red(lo_x, np.min, 16)
    th = red(hi_y, np.max, 16) - red(lo_y, np.min, 16)
    gw, gh, tw, th = [np.maximum(a, 0) for a in (gw, gh, tw, th)]
    pitch = (tw + 3 + 3) // 4 * 4   # up to 3 pixels of start alignment
    n = tw.size
    print(f"r{r} G{G} {H}x{W}: {n} tiles; group cols > 32: {np.mean((gw > 32).reshape(B, ty, tx, 2).any(3)):.3f}  group rows > 20: "
          f"{np.mean((gh > 20).reshape(B, ty, tx, 2).any(3)):.3f} (> 24: {np.mean((gh > 24).reshape(B, ty, tx, 2).any(3)):.3f})  "
          f"positions > 1038: {np.mean(pitch * th > 1038):.3f} (> 1600: {np.mean(pitch * th > 1600):.3f}, > 2200: {np.mean(pitch * th > 2200):.3f})")
    print("   group width percentiles 50/90/99:", np.percentile(gw, [50, 90, 99]), " group height:", np.percentile(gh, [50, 90, 99]),
          " tile positions:", np.percentile(pitch * th, [50, 90, 99]))
