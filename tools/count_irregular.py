"""How many tiles of each local-correlation call of the bench step go to the irregular (second) launch."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gfnet_amd import _lib  # noqa: E402

L = _lib.lib()
orig = L.gfn_local_corr_fwd
seen = []


def wrapped(*a):
    rc = orig(*a)
    torch.cuda.synchronize()
    B, C, G, H, W, r = (int(getattr(v, "value", v)) for v in a[7:13])
    scratch = a[16]
    n = ctypes.cast(scratch, ctypes.POINTER(ctypes.c_int))
    cnt = torch.empty(1, dtype=torch.int32)
    import ctypes as ct
    hip = ct.CDLL("libamdhip64.so")
    buf = ct.c_int(0)
    hip.hipMemcpy(ct.byref(buf), ct.c_void_p(getattr(scratch, 'value', scratch) + 12), 4, 2)  # int 3: the last call's count
    rounds = 2 if r <= 4 else 1
    tiles = B * ((G + 15) // 16) * ((G + 2 * rounds - 1) // (2 * rounds))
    seen.append((C, H, G, r, buf.value, tiles))
    return rc


L.gfn_local_corr_fwd = wrapped
sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--cpu-pairs", "0"]
bench.main()
for C, H, G, r, cnt, tiles in seen[:7]:
    print(f"c{C} hs{H} G{G} r{r}: {cnt} of {tiles} tiles irregular ({100.0*cnt/tiles:.1f} %)")
