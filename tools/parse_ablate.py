"""Group a rocprofv3 kernel-trace CSV of tools/ablate_local_corr.py into per-(shape, mask) mean kernel durations."""
import csv
import glob
import sys

sys.path.insert(0, "tools")
SHAPES = [(64, 32, 32, 7), (32, 112, 64, 4), (16, 224, 128, 2)]
MASKS = [0, 1, 2, 4, 8, 16, 32, 1 | 2, 1 | 2 | 4, 1 | 2 | 4 | 8 | 16 | 32, 2 | 4 | 8 | 16 | 32, 1 | 4 | 8 | 16 | 32, 16 | 32]
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(path)) if "local_corr_tile" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = 12
i = 0
for sh in SHAPES:
    for m in MASKS:
        grp = rows[i:i + per][2:]
        i += per
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in grp]
        names = [n for b, n in ((1, "stage"), (2, "dstage"), (4, "f0"), (8, "epi"), (16, "fallback"), (32, "table")) if m & b]
        print(f"{sh}: skip[{'+'.join(names) or 'nothing':38s}] mean {sum(d)/len(d):8.1f} us  min {min(d):8.1f}")
print("VGPR", rows[0].get("VGPR_Count"), "LDS", rows[0].get("LDS_Block_Size"), "scratch", rows[0].get("Scratch_Size"))
