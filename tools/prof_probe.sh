cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02_kt -o kt --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/probe_local_corr.py 64 homography > $GRAFT_REPO_ROOT/gpurun_out/r02_kt.log 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/r02_kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:90], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
