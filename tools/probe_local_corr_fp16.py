import sys, os
import torch
sys.path.insert(0, os.getcwd())
from gfnet_amd.utils.local_correlation import local_correlation
for dt in (torch.float32, torch.float16):
    for kind in ("random", "smooth"):
        for (c, hs, G, r, B) in [(64, 48, 48, 7, 16), (64, 32, 32, 7, 16), (64, 84, 48, 6, 16)]:
            f0 = torch.randn(B, c, G, G, device="cuda")
            f1 = torch.randn(B, c, hs, hs, device="cuda").to(dt)
            if kind == "random":
                flow = torch.rand(B, 2, G, G, device="cuda") * 1.8 - 0.9
            else:
                lin = torch.linspace(-0.9, 0.9, G, device="cuda")
                gy, gx = torch.meshgrid(lin, lin, indexing="ij")
                flow = torch.stack((gx, gy))[None].repeat(B, 1, 1, 1).contiguous()
            out = torch.empty(B, (2 * r + 1) ** 2, G, G, device="cuda")
            for _ in range(3):
                local_correlation((B, c, hs, hs), f0, f1, r, G, flow=flow, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                local_correlation((B, c, hs, hs), f0, f1, r, G, flow=flow, out=out)
            e1.record(); torch.cuda.synchronize()
            print(dt, kind, (c, hs, G, r, B), f"{e0.elapsed_time(e1) * 100:.1f} us", flush=True)
