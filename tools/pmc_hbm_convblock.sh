#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_hbm_convblock.sh TAG C G [B]  -> FETCH_SIZE / WRITE_SIZE / L2 hit rate per dispatch
TAG=$1; C=$2; G=$3; B=${4:-64}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/hbm_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set -d $OUT/p$i -o p$i --output-format csv -- python3 $ROOT/tools/probe_convblock.py $C $G $B ${PMC_VARIANT:-0} 2 > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "fused" in k or "dw5x5" in k or "pw_gemm" in k:
        print(k)
        for c, v in sorted(d.items()): print(f"   {c:28s} per dispatch {sum(v)/len(v):.5g}  (n={len(v)})")
PY
