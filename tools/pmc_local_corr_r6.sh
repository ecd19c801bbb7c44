#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_local_corr_r6.sh C HS G R TAG [LIB]  -> the round-6 subset of the SQ counters of the lean tile kernel
# (separate --pmc passes, no trace domains), per dispatch and per wave; LIB = a variant library (GFNET_HIP_LIB)
C=$1; HS=$2; G=$3; R=$4; TAG=${5:-cur}; LIB=$6
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc6_r${R}_$TAG
mkdir -p $OUT
if [ -n "$LIB" ]; then export GFNET_HIP_LIB=$ROOT/$LIB; fi
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY" \
           "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32" \
           "SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set -d $OUT/p$i -o p$i --output-format csv -- python3 $ROOT/tools/probe_local_corr_one.py $C $HS $G $R 64 4 > $OUT/p$i.log 2>&1
done
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "local_corr" in k:
        print(k)
        w = sum(d["SQ_WAVES"]) / len(d["SQ_WAVES"]) if "SQ_WAVES" in d else 1
        for c, v in sorted(d.items()): print(f"   {c:32s} per dispatch {sum(v)/len(v):12.5g}  per wave {sum(v)/len(v)/w:10.5g} (n={len(v)})")
PY
rm -rf $OUT/p*/
cat $OUT/summary.txt
