#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_local_corr_r5.sh C HS G R [TAG]  -> SQ / SQC counters of the lean tile kernel, per dispatch and per
# wave, instruction-fetch and instruction-cache counters included (round 5; separate --pmc passes, no trace domains)
C=$1; HS=$2; G=$3; R=$4; TAG=${5:-cur}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc5_r${R}_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY" \
           "SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_BUSY_CU_CYCLES" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS SQ_ACTIVE_INST_ANY" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQC_DCACHE_REQ SQC_DCACHE_MISSES" \
           "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32" \
           "SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES"; do
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
rm -rf $OUT/p*/   # keep the summary and logs only
cat $OUT/summary.txt
