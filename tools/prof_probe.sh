#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_probe.sh [NAME]  -> rocprofv3 kernel stats of tools/probe_local_corr.py (64 directions,
# homography flows, every production shape) into gpurun_out/NAME (default kt_probe)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=${1:-kt_probe}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/$NAME -o kt --output-format csv -- python3 $ROOT/tools/probe_local_corr.py 64 homography > $ROOT/gpurun_out/$NAME.log 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob("$ROOT/gpurun_out/$NAME/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:90], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
