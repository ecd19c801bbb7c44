#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_local_corr.sh C HS G R  -> SQ counters of the tile kernel, per dispatch
C=$1; HS=$2; G=$3; R=$4
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_lc_r${R}_v${VARIANT:-0}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY" \
           "SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set -d $OUT/p$i -o p$i --output-format csv -- python3 $ROOT/tools/probe_local_corr_one.py $C $HS $G $R 64 4 > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "local_corr_tile" in k:
        print(k)
        w = sum(d["SQ_WAVES"]) / len(d["SQ_WAVES"]) if "SQ_WAVES" in d else 1
        for c, v in sorted(d.items()): print(f"   {c:28s} per dispatch {sum(v)/len(v):.5g}  per wave {sum(v)/len(v)/w:.5g} (n={len(v)})")
PY
