"""stdin: 'CS wg wave index ticks' lines of tools/stamp_convblock.py -> one row per (workgroup, wave): phase durations in ticks (100 MHz)."""
import sys, collections
rows = collections.defaultdict(dict)
for line in sys.stdin:
    f = line.split()
    if len(f) == 5 and f[0] == "CS":
        rows[(int(f[1]), int(f[2]))][int(f[3])] = int(f[4])
for key in sorted(rows)[:int(sys.argv[1]) if len(sys.argv) > 1 else 8]:
    st = rows[key]
    idx = sorted(st)
    prev = 0
    parts = []
    for i in idx:
        parts.append(f"{i}:+{st[i] - prev}")
        prev = st[i]
    print(key, "total", st[idx[-1]], " ".join(parts))
