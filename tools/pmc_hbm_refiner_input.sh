#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_hbm_refiner_input.sh C HS G DD -> FETCH_SIZE / WRITE_SIZE per dispatch of refiner_input_kernel
C=$1; HS=$2; G=$3; DD=$4
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/hbm_ri_$G
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set -d $OUT/p$i -o p$i --output-format csv -- python3 $ROOT/tools/probe_refiner_input.py $C $HS $G $DD 5 > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "refiner_input" in k:
        print(k)
        for c, v in sorted(d.items()): print(f"   {c:28s} per dispatch {sum(v)/len(v):.5g}  (n={len(v)})")
PY
