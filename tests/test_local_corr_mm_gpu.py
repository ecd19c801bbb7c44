"""GPU: the parked r = 3 / 4 matrix-core local-correlation kernel (csrc/local_corr_mw.h on the helpers of csrc/local_corr_mm.h, round 3)
stays parity-green.  It is not the product path (the lean fp32 kernel is faster, profiles/r03_local_corr_mm.md); `gfnet_amd/build.py
--mm` (or __graft_entry__.build() with GFN_BUILD_MM=1; opt-in since round 4) compile it into libgfnet_hip_mm.so (-DGFN_MM_DEFAULT=1: default path of r = 3, 4), which a child
process loads through GFNET_HIP_LIB.  (The r >= 5 matrix-core kernel IS the product path: tests/test_local_corr_gpu.py.)
Split-bf16 products: NOT bit-identical to the fp32 FMA kernels, within 1e-4 * max(1, |ref|) of the oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MM_LIB = os.path.join(ROOT, "gfnet_amd", "csrc", "libgfnet_hip_mm.so")

CHILD = r"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["GFN_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GFN_ROOT"], "tests", "golden"))
import oracle, synth
from gfnet_amd import _lib
from gfnet_amd.utils.local_correlation import local_correlation
assert _lib.LIB_PATH.endswith("libgfnet_hip_mm.so"), _lib.LIB_PATH
worst = 0.0
for (c, hs, G, r) in [(32, 112, 64, 4), (32, 140, 80, 4), (32, 56, 32, 4), (16, 60, 40, 3)]:
    for kind in ("homography", "zoom", "border", "random"):
        B = 2
        f0 = synth.lattice_normalish((B, c, G, G), 31 + r)
        f1 = synth.lattice_normalish((B, c, hs, hs), 32 + r)
        if kind == "homography":
            flow = synth.homography_flow(B, G, 33)
        elif kind == "zoom":
            flow = synth.homography_flow(B, G, 34, scale=1.35)
        elif kind == "border":
            flow = synth.homography_flow(B, G, 36, scale=1.02)
            flow[0, 0] += np.float32(0.35); flow[1, 1] -= np.float32(0.4)
        else:
            flow = 1.2 * synth.lattice_uniform((B, 2, G, G), 35)
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        args = ((B, c, hs, hs), dev(f0), dev(f1), r, G)
        mm = local_correlation(*args, flow=dev(flow)).cpu().numpy()                  # variant 0: the matrix-core kernel in this build
        lean = local_correlation(*args, flow=dev(flow), _variant=4).cpu().numpy()    # fp32 FMA lean kernel (r <= 4) / round-1 kernel
        old = local_correlation(*args, flow=dev(flow), _variant=2).cpu().numpy()
        ref = oracle.local_correlation((B, c, hs, hs), f0, f1, r, G, flow=flow)
        err = float((np.abs(mm - ref) / np.maximum(1, np.abs(ref))).max())
        worst = max(worst, err)
        assert err < 1e-4, (c, hs, G, r, kind, err)
        assert np.array_equal(lean, old), (c, hs, G, r, kind, "variant 4 vs 2")
        if kind == "homography" and r <= 4:
            assert not np.array_equal(mm, lean), "variant 0 of this build must be the matrix-core kernel"
print("MM_PARITY_OK worst", worst)
"""


def test_matrix_core_kernel_matches_the_oracle():
    if not os.path.exists(MM_LIB):  # opt-in build (a second four-minute compile for a kernel that is not the product path)
        pytest.skip("libgfnet_hip_mm.so not built: `python -m gfnet_amd.build --mm` or GFN_BUILD_MM=1 with __graft_entry__.build()")
    env = dict(os.environ, GFNET_HIP_LIB=MM_LIB, GFN_ROOT=ROOT)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "MM_PARITY_OK" in r.stdout
