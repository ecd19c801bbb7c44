"""Timing probe (GPU): the refiners' conv stacks (torch / MIOpen) on the bench shapes."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gfnet_amd.model.network import _refiner_for

shapes448 = [("16", 64, 64, 7, 32), ("8", 64, 64, 6, 32), ("4", 32, 32, 4, 64), ("2", 16, 16, 2, 128), ("1", 8, 8, 0, 256)]
shapes560 = [("8", 64, 64, 6, 40), ("4", 32, 32, 4, 80), ("2", 16, 16, 2, 160), ("1", 8, 8, 0, 320)]
B = 64
tot = {}
for amp in (True, False):
    total = 0.0
    for name, shapes in (("448", shapes448), ("560", shapes560)):
        for (s, feat, disp, r, G) in shapes:
            ref = _refiner_for(feat, disp, r).cuda().eval()
            ref.amp = amp
            dim = ref.block1[0].in_channels
            d = torch.randn(B, dim, G, G, device="cuda")
            def run():
                with torch.no_grad(), torch.autocast("cuda", enabled=amp, dtype=torch.float16):
                    h = ref.hidden_blocks(ref.block1(d))
                return ref.out_conv(h.float())
            for _ in range(2): run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): run()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3
            total += ms
            print(f"amp={amp} pass {name} scale {s}: C={dim} G={G}: {ms:.2f} ms", flush=True)
    print(f"amp={amp}: conv stacks total per step (64 directions, 448+560): {total:.1f} ms")
