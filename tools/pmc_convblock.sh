#!/bin/bash
# usage (on the GPU box, from repo root): bash tools/pmc_convblock.sh TAG C G [B]
# separate --pmc passes (never combined with trace domains), outputs under gpurun_out/pmc_TAG/
TAG=$1; C=$2; G=$3; B=${4:-64}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_SALU" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VALU_FMA_F32 SQ_LDS_ADDR_CONFLICT" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY" \
           "TA_BUSY_sum TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_LEVEL_WAVES SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set -d $OUT/p$i -o p$i --output-format csv -- python3 $ROOT/tools/probe_convblock.py $C $G $B ${PMC_VARIANT:-0} 2 > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in agg.items():
    if "fused" in k or "dw5x5" in k or "pw_gemm" in k:
        print(k)
        for c, v in sorted(d.items()): print(f"   {c:32s} {v:.4g}")
PY
