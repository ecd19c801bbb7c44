"""How the plan launch routes the tiles of a lean local-correlation call (GPU): python tools/plan_flags.py [homography|bench]
Reads the plan records (16 ints per tile, word 3 = flags) back from the scratch buffer after one call."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import synth  # noqa: E402
from gfnet_amd import _lib  # noqa: E402
from gfnet_amd import _synthetic as synthetic  # noqa: E402
from gfnet_amd.utils.local_correlation import local_correlation  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "bench"
B = 64
for (c, hs, G, r) in [(32, 112, 64, 4), (32, 140, 80, 4), (16, 224, 128, 2), (16, 280, 160, 2)]:
    g = torch.Generator().manual_seed(1)
    f0 = torch.randn(B, c, G, G, generator=g).cuda()
    f1 = torch.randn(B, c, hs, hs, generator=g).cuda()
    if kind == "homography":
        flow = torch.from_numpy(np.tile(synth.homography_flow(2, G, 5), (B // 2, 1, 1, 1))).cuda()
    else:
        S = 4 * hs if r == 4 else 2 * hs
        Hm = synthetic.random_homographies(B // 2, S, g)
        flow = torch.cat((synthetic.warp_grid(Hm, G, S, "cpu"), synthetic.warp_grid(np.linalg.inv(Hm), G, S, "cpu"))).permute(0, 3, 1, 2)
        flow = (flow + torch.randn(B, 2, G, G, generator=g) * (0.5 / S)).contiguous().cuda()
    out = torch.empty(B, (2 * r + 1) ** 2, G, G, device="cuda")
    local_correlation((B, c, hs, hs), f0, f1, r, G, flow=flow, out=out)
    torch.cuda.synchronize()
    nbytes = int(_lib.lib().gfn_local_corr_scratch_bytes(B, G))
    scratch = _lib.scratch(torch.device("cuda", 0), nbytes)
    raw = scratch.view(torch.uint8)[:nbytes].cpu().numpy()
    tiles_max = ((G + 1) // 2) * ((G + 15) // 16) * B
    base = scratch.data_ptr()
    off = ((base + 4 * (tiles_max + 8) + 31) & ~31) - base
    tiles = B * ((G + 3) // 4) * ((G + 15) // 16)
    plan = raw[off:off + tiles * 64].view(np.int32).reshape(tiles, 16)
    fl = plan[:, 3]
    w, h = plan[:, 2] & 0xffff, plan[:, 2] >> 16
    pitch = plan[:, 7] & 0xff
    print(f"{kind} c{c} {hs}^2 G{G} r{r}: tiles {tiles}; halves {np.mean((fl & 4) != 0):.3f}, second {np.mean((fl & 2) != 0):.4f}, "
          f"interior {np.mean((fl & 1) != 0):.3f}; region w x h median {int(np.median(w))} x {int(np.median(h))} (max {w.max()} x {h.max()}), pitch median {int(np.median(pitch))}", flush=True)
