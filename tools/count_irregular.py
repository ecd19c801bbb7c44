"""Per local-correlation call of one bench step: tiles left to the second launch, cells redone tap by tap, tiles staged in two
halves -- the counters the kernels leave in the scratch header (csrc/local_corr.hip kTodoHdr), read through ops.kernel_counters.
usage (GPU box): python tools/count_irregular.py [bench args, e.g. --workload 672b16]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gfnet_amd import ops  # noqa: E402

sys.argv = ["bench.py", "--steps", "1", "--warmup", "1", "--cpu-pairs", "0"] + sys.argv[1:]
bench.main()
ops.kernel_counters = {}
real_stdout, sys.stdout = sys.stdout, open(os.devnull, "w")
try:
    bench.main()  # one more step with the counters collected (a device sync per call)
finally:
    sys.stdout = real_stdout
for name, rows in ops.kernel_counters.items():
    # name = local_corr_c{C}_h{Hs}_g{G}_r{r}
    f = dict((p[0], int(p[1:])) for p in name.split("_")[2:])
    rounds = 2 if f["r"] <= 4 else 1
    tiles_per_dir = ((f["g"] + 15) // 16) * ((f["g"] + 2 * rounds - 1) // (2 * rounds))
    n = len(rows)
    second = sum(r[0] for r in rows) / n
    flagged = sum(r[1] for r in rows) / n
    halves = sum(r[2] for r in rows) / n
    print(f"{name}: {n} calls; per call {second:.1f} tiles to the second launch, {flagged:.1f} cells redone per tap, "
          f"{halves:.1f} (sampled) tiles staged in halves; {tiles_per_dir} tiles per direction")
