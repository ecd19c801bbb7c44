"""Phase time stamps of one workgroup of the tiled local-correlation kernel (GFN_ABLATE build, device printf)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["GFNET_HIP_LIB"] = os.path.join(ROOT, "gfnet_amd", "csrc", "libgfnet_hip_ablate.so")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import synth  # noqa: E402
from gfnet_amd.utils.local_correlation import local_correlation  # noqa: E402

B, c, hs, G, r = (64, 32, 112, 64, 4) if len(sys.argv) < 2 else tuple(int(v) for v in sys.argv[1:6])
f0 = torch.randn(B, c, G, G, device="cuda")
f1 = torch.randn(B, c, hs, hs, device="cuda")
flow = torch.from_numpy(np.tile(synth.homography_flow(2, G, 5), (B // 2, 1, 1, 1))).cuda()
out = torch.empty(B, (2 * r + 1) ** 2, G, G, device="cuda")
for _ in range(3):
    local_correlation((B, c, hs, hs), f0, f1, r, G, flow=flow, out=out)
torch.cuda.synchronize()
for _ in range(3):
    local_correlation((B, c, hs, hs), f0, f1, r, G, flow=flow, out=out, _variant=512 << 8)
    torch.cuda.synchronize()
