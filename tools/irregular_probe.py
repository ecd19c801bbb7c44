"""Where does the second (irregular) local-correlation launch spend its time on the bench's own flows?
Captures the refiner inputs of one bench step and re-times their local-correlation call with stages ablated
(GFN_ABLATE build: bit 16 = no flagged-cell redo, bit 1024 = no gather variant)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["GFNET_HIP_LIB"] = os.path.join(ROOT, "gfnet_amd", "csrc", "libgfnet_hip_ablate.so")
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from gfnet_amd import _lib, ops  # noqa: E402
from gfnet_amd._lib import c_vp, ptr, stream_ptr  # noqa: E402

calls = []
orig = ops.refiner_input


def spy(num_grid, x, y, flow, disp_w, disp_b, local_radius, scale_factor=1.0, corr_in_other=True):
    d = orig(num_grid, x, y, flow, disp_w, disp_b, local_radius, scale_factor, corr_in_other)
    if corr_in_other and len(calls) < 4:
        calls.append((int(num_grid), x.clone(), y.clone(), flow.clone(), int(local_radius), d.clone(), disp_w.shape[0]))
    return d


ops.refiner_input = spy
sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--cpu-pairs", "0"]
bench.main()
L = _lib.lib()


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for G, x, y, fl, r, d, Dd in calls:
    Bi, C, Hs, Ws = x.shape
    B = fl.shape[0]
    CH = d.shape[1]
    out = d[:, 2 * C + Dd:]
    nscr = int(L.gfn_local_corr_scratch_bytes(B, G))
    scr = torch.zeros(nscr, dtype=torch.uint8, device="cuda")
    st = stream_ptr(x.device)
    print(f"c{C} hs{Hs} G{G} r{r}")
    for name, v in (("full", 0), ("no flagged-cell redo", 16), ("no gather variant", 1024), ("neither", 1040), ("gather loads at offset 0", 2048)):
        def run():
            rc = L.gfn_local_corr_fwd_ex(ptr(d), CH * G * G, ptr(y), ptr(x), ptr(fl), c_vp(out.data_ptr()), CH * G * G, B, C, G, Hs, Ws,
                                         r, 0, Hs, Ws, v << 8, ptr(scr), nscr, st)
            assert rc == 0, rc
        print(f"   {name:24s} {timeit(run):8.1f} us")
