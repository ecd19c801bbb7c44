"""stdin: 'CS wg wave index ticks' lines of tools/stamp_convblock.py -> per phase the mean duration (shader cycles) over the stamped
workgroups' waves, and the rows themselves with --rows N."""
import sys, collections
rows = collections.defaultdict(dict)
for line in sys.stdin:
    f = line.split()
    if len(f) == 5 and f[0] == "CS":
        rows[(int(f[1]), int(f[2]))][int(f[3])] = int(f[4])
names = {1: "first issue", 2: "zero barrier"}
for t in range(5):
    names.update({3 + 6 * t: f"commit{t}", 4 + 6 * t: f"barrier{t}a", 5 + 6 * t: f"issue{t}", 6 + 6 * t: f"dw{t}", 7 + 6 * t: f"barrier{t}b", 8 + 6 * t: f"mfma{t}"})
agg = collections.defaultdict(list)
tot = []
late = any(k[0] >= 1000 for k in rows)
for key, st in rows.items():
    if late and key[0] < 1000:
        continue  # first round of workgroups: cold instruction cache
    idx = [i for i in sorted(st) if i < 33]
    prev = 0
    for i in idx:
        agg[names.get(i, str(i))].append(st[i] - prev)
        prev = st[i]
    tot.append(max(st.values()))
    ends = sorted(st[i] for i in st if i >= 33)
    if ends:
        agg["stores(last)"].append(ends[-1] - st[idx[-1]])
print("workgroup-waves", len(tot), "mean total", sum(tot) // max(len(tot), 1))
print(" ".join(f"{k}={sum(v) // len(v)}" for k, v in agg.items()))
cls = collections.defaultdict(int)
for k, v in agg.items():
    base = "".join(c for c in k if not c.isdigit())
    cls[base] += sum(v) // len(v)
print("by phase:", " ".join(f"{k}={v}" for k, v in cls.items()))
