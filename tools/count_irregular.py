"""Per local-correlation call of one bench step: tiles left to the second launch, cells redone tap by tap, tiles staged in two
halves (sampled) -- the counters the kernels leave in the scratch header (csrc/local_corr.hip kTodoHdr), which bench.py
--breakdown collects through ops.kernel_counters.
usage (GPU box): python tools/count_irregular.py [bench args, e.g. --workload 672b16]"""
import ast
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-pairs", "0", "--breakdown", "--no-stack-leg"] + sys.argv[1:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
for line in err.splitlines():
    if not line.startswith("[counters] "):
        continue
    name, rest = line[len("[counters] "):].split(":", 1)
    rows = ast.literal_eval(rest.split(":", 1)[1].strip())
    f = dict((p[0], int(p[1:])) for p in name.split("_")[2:])  # local_corr_c{C}_h{Hs}_g{G}_r{r}
    rounds = 2 if f["r"] <= 4 else 1
    tiles = ((f["g"] + 15) // 16) * ((f["g"] + 2 * rounds - 1) // (2 * rounds))
    n = len(rows)
    print(f"{name}: {n} call(s) per step; per call {sum(r[0] for r in rows) / n:.1f} tiles to the second launch, "
          f"{sum(r[1] for r in rows) / n:.1f} cells redone per tap, {sum(r[2] for r in rows) / n:.1f} tiles staged in halves (sampled); "
          f"{tiles} tiles per direction")
