#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_lds_conflicts.sh C HS G R  -> SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE per tile wave of the lean tile kernel
# with one phase switched off at a time (GFN_ABLATE build: python -m gfnet_amd.build --ablate); says which phase the bank conflicts are in
C=$1; HS=$2; G=$3; R=$4
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ldsconf_r$R
mkdir -p $OUT
export GFNET_HIP_LIB=$ROOT/gfnet_amd/csrc/libgfnet_hip_ablate.so
cd /tmp && export TMPDIR=/tmp
for m in 0 1 2 4 8 16 32; do
  VARIANT=$((m << 8)) timeout 120 rocprofv3 --pmc SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d $OUT/m$m -o m$m --output-format csv -- python3 $ROOT/tools/probe_local_corr_one.py $C $HS $G $R 64 4 > $OUT/m$m.log 2>&1
done
python3 - <<PY
import csv, glob, collections
names = {0: "as built", 1: "no staging", 2: "no D-stage", 4: "no f0 block", 8: "no epilogue (blend + stores)", 16: "no D-buffer writes", 32: "no fraction table"}
for m in (0, 1, 2, 4, 8, 16, 32):
    agg = collections.defaultdict(list)
    for f in glob.glob("$OUT/m%d/**/*counter_collection.csv" % m, recursive=True):
        for r in csv.DictReader(open(f)):
            if "tile2" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    w = sum(agg["SQ_WAVES"]) / max(len(agg["SQ_WAVES"]), 1)
    g = lambda k: sum(agg[k]) / max(len(agg[k]), 1) / max(w, 1)
    print(f"{names[m]:32s} conflicts {g('SQ_LDS_BANK_CONFLICT'):7.1f}  LDS active {g('SQ_LDS_IDX_ACTIVE'):7.1f}  LDS instructions {g('SQ_INSTS_LDS'):6.1f}  (per wave)")
PY
rm -rf $OUT/m*/
