import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import oracle
from gfnet_amd import ops
torch.manual_seed(0)
for (n, amp) in ((32, 1.0), (32, 6.0), (48, 2.0)):
    f0 = (torch.randn(2, 64, n, n) * amp)
    f1 = (torch.nn.functional.avg_pool2d(torch.randn(2, 64, n, n), 3, 1, 1) * amp * 3)
    ref64 = oracle.corr_softargmax(f0.numpy(), f1.numpy(), variant="f64")
    ref32 = oracle.corr_softargmax(f0.numpy(), f1.numpy())
    got = ops.corr_softargmax(f0.cuda(), f1.cuda(), symmetric=False).cpu().numpy()
    print(f"{n}x{n} amp {amp}: |hip - f64 oracle| max {np.abs(got - ref64).max():.3e}   |f32 oracle - f64 oracle| max {np.abs(ref32 - ref64).max():.3e}")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a = torch.randn(32, 64, n, n, device="cuda"); b = torch.randn(32, 64, n, n, device="cuda")
    for _ in range(3): ops.corr_softargmax(a, b, symmetric=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): ops.corr_softargmax(a, b, symmetric=True)
    e1.record(); torch.cuda.synchronize()
    print(f"   64 directions: {e0.elapsed_time(e1) * 50:.1f} us per call")
