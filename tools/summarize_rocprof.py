"""Turn a rocprofv3 --kernel-trace --stats output directory into a small committed summary
(profiles/<name>.md + the raw kernel_stats.csv)."""
import csv
import glob
import os
import shutil
import sys

src, name = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
stats = glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(stats)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
os.makedirs("profiles", exist_ok=True)
shutil.copy(stats, os.path.join("profiles", name + "_kernel_stats.csv"))
with open(os.path.join("profiles", name + ".md"), "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats : {name}\n\n{note}\n\n")
    f.write(f"Total kernel time {tot/1e6:.3f} ms over the profiled process.\n\n")
    f.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
    for r in rows[:40]:
        f.write(f"| `{r['Name'][:110]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | "
                f"{float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |\n")
print(open(os.path.join("profiles", name + ".md")).read()[:1500])
