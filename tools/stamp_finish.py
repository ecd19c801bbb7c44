"""Phase time stamps of one workgroup of the homography finish kernel (GFN_ABLATE build, device printf)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["GFNET_HIP_LIB"] = os.path.join(ROOT, "gfnet_amd", "csrc", "libgfnet_hip_ablate.so")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from gfnet_amd import ops  # noqa: E402
from test_homography_cpu import make_points, random_h  # noqa: E402

rng = np.random.default_rng(0)
pts = torch.from_numpy(np.stack([make_points(rng, random_h(rng), 5000, noise=0.25, outliers=0.02) for _ in range(32)])).cuda()
for _ in range(2):
    ops.find_homography(pts, iters=2000)
    torch.cuda.synchronize()
