"""Print per-kernel stats from a rocprofv3 --kernel-trace --stats output dir: python tools/kstats.py DIR [substring]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(f)):
    if sub in r["Name"]:
        print(f"{r['Name'][:100]:100s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f}  max {float(r['MaxNs'])/1e3:9.1f}")
