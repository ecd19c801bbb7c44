"""GPU probe: one refiner conv block (csrc/conv_stack.hip) on one shape, for rocprofv3 runs.
usage: python tools/probe_convblock.py C G [B] [variant] [reps]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gfnet_amd import ops

C, G = int(sys.argv[1]), int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
variant = int(sys.argv[4]) if len(sys.argv) > 4 else 0
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
torch.manual_seed(0)
x = torch.randn(B, C, G, G, device="cuda")
packed = ops.conv_block_pack(torch.randn(C, 25, device="cuda") * 0.2, torch.randn(C, device="cuda"), torch.rand(C, device="cuda") + 0.5,
                             torch.randn(C, device="cuda"), torch.randn(C, C, device="cuda") * C ** -0.5, torch.randn(C, device="cuda"))
y = torch.empty_like(x)
t = torch.empty_like(x) if variant == 1 else None
for _ in range(2):
    ops.conv_block(x, packed, C, out=y, variant=variant, t_scratch=t)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.conv_block(x, packed, C, out=y, variant=variant, t_scratch=t)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
print(f"conv_block C={C} G={G} B={B} variant={variant}: {us:.1f} us  {2.0*B*C*C*G*G/us/1e6:.1f} TFLOP/s  {2*B*C*G*G*4/us/1e6:.2f} TB/s (x+y)")
