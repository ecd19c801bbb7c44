"""Busy time of a rocprofv3 kernel trace: python tools/trace_overlap.py DIR [tail_fraction]
   -> over the last part of the trace: wall span, union of kernel intervals (time with at least one kernel running), sum of durations,
      the same per stream / queue, and the kernels ranked by their share of the span in which they run ALONE."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
t_end = max(e[1] for e in ev)
t_begin = min(e[0] for e in ev)
cut = t_end - (t_end - t_begin) * frac
ev = [e for e in ev if e[0] >= cut]
span = ev[-1][1] - ev[0][0]
total = sum(e[1] - e[0] for e in ev)
# union + alone time by sweep
pts = []
for i, (a, b, n, q) in enumerate(ev):
    pts.append((a, 1, i))
    pts.append((b, -1, i))
pts.sort()
active = set()
union = 0
alone = collections.Counter()
last = pts[0][0]
for t, kind, i in pts:
    if active:
        union += t - last
        if len(active) == 1:
            alone[ev[next(iter(active))][2][:70]] += t - last
    last = t
    if kind == 1:
        active.add(i)
    else:
        active.discard(i)
print(f"{len(ev)} kernels over {span / 1e6:.3f} ms: busy (>= 1 kernel) {union / 1e6:.3f} ms = {union / span:.1%}, sum of durations {total / 1e6:.3f} ms = {total / span:.2f}x")
per_q = collections.defaultdict(int)
for a, b, n, q in ev:
    per_q[q] += b - a
for q, v in sorted(per_q.items(), key=lambda kv: -kv[1]):
    print(f"  queue {q}: {v / 1e6:.3f} ms of kernels = {v / span:.1%} of the span")
print("running alone (share of the span):")
for n, v in alone.most_common(14):
    print(f"  {v / span:6.1%}  {n}")
