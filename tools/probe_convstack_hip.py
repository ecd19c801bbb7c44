"""Timing probe (GPU): refiner conv stacks on the bench shapes -- HIP (csrc/conv_stack.hip) per kernel."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gfnet_amd import ops
from gfnet_amd.model.network import _refiner_for

shapes448 = [("16", 64, 64, 7, 32), ("8", 64, 64, 6, 32), ("4", 32, 32, 4, 64), ("2", 16, 16, 2, 128), ("1", 8, 8, 0, 256)]
shapes560 = [("8", 64, 64, 6, 40), ("4", 32, 32, 4, 80), ("2", 16, 16, 2, 160), ("1", 8, 8, 0, 320)]
B = int(os.environ.get("PROBE_B", 64))


def timeit(fn, n=3):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for variant in [int(v) for v in os.environ.get("PROBE_VARIANTS", "0,2").split(",")]:
    total = 0.0
    for name, shapes in (("448", shapes448), ("560", shapes560)):
        for (s, feat, disp, r, G) in shapes:
            ref = _refiner_for(feat, disp, r).cuda().eval()
            C = ref.block1[0].in_channels
            d = torch.randn(B, C, G, G, device="cuda")
            with torch.no_grad():
                ms = timeit(lambda: ref.conv_stack(d, variant=variant))
                fold, _ = ref.folded_stack()
                y = torch.empty_like(d)
                t = torch.empty_like(d)
                ms_blk = timeit(lambda: ops.conv_block(d, fold[0][0], C, out=y, variant=variant, t_scratch=t))
            total += ms
            byts = B * C * G * G * 4
            fl = 2.0 * B * C * C * G * G
            print(f"variant={variant} pass {name} scale {s}: C={C} G={G}: stack {ms:.2f} ms | block {ms_blk*1e3:.0f} us "
                  f"({fl/ms_blk/1e9:.1f} TFLOP/s, {2*byts/ms_blk/1e9:.2f} TB/s x+y)", flush=True)
    print(f"variant={variant}: HIP conv stacks total per step ({B} directions, 448+560): {total:.1f} ms", flush=True)
