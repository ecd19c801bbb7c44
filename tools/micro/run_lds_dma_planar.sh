#!/bin/bash
# usage (GPU box, repo root): bash tools/micro/run_lds_dma_planar.sh  -> timings + SQ counters per wave of every mode of tools/micro/lds_dma_planar
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ldsdma
mkdir -p $OUT
$ROOT/tools/micro/lds_dma_planar | tee $OUT/timing.txt
cd /tmp && export TMPDIR=/tmp
for m in 0 1 2 3; do
  timeout 120 rocprofv3 --pmc SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU -d $OUT/m$m -o m$m --output-format csv -- $ROOT/tools/micro/lds_dma_planar $m > $OUT/m$m.log 2>&1
  timeout 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -d $OUT/n$m -o n$m --output-format csv -- $ROOT/tools/micro/lds_dma_planar $m > $OUT/n$m.log 2>&1
done
python3 - <<PY | tee $OUT/counters.txt
import csv, glob, collections
for m in range(4):
    agg = collections.defaultdict(list)
    for f in glob.glob("$OUT/[mn]%d/**/*counter_collection.csv" % m, recursive=True):
        for r in csv.DictReader(open(f)):
            if "tile_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    w = sum(agg["SQ_WAVES"]) / max(len(agg["SQ_WAVES"]), 1)
    g = lambda k: sum(agg[k]) / max(len(agg[k]), 1) / max(w, 1) / 8.0   # per wave and TILE (a wave walks 8 tiles)
    print(f"MODE {m}: per tile wave: LDS active {g('SQ_LDS_IDX_ACTIVE'):7.1f} clocks (bank conflicts {g('SQ_LDS_BANK_CONFLICT'):6.1f}), LDS instructions {g('SQ_INSTS_LDS'):6.1f}, "
          f"VALU instructions {g('SQ_INSTS_VALU'):7.1f}, wave cycles {4 * g('SQ_WAVE_CYCLES'):8.0f}")
PY
rm -rf $OUT/m*/ $OUT/n*/
