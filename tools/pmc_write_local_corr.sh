#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_write_local_corr.sh C HS G R -> the L2's memory-side WRITE request counters of the
# local-correlation tile kernel, for the product build (streaming "nt" stores of the correlation planes) and the plain-store build
# (libgfnet_hip_plainst.so = local_corr.hip with -DGFN_LEAN_ST_AUX=0): settles whether WRITE_SIZE's 1.30x on the nt-stored planes is
# real traffic (more 32-byte / partial requests) or an accounting artefact.  One --pmc pass per counter pair.
C=$1; HS=$2; G=$3; R=$4
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/wr_lc_r$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for build in nt plain; do
  if [ $build = plain ]; then export GFNET_HIP_LIB=$ROOT/gfnet_amd/csrc/libgfnet_hip_plainst.so; else unset GFNET_HIP_LIB; fi
  i=0
  for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "WRITE_SIZE" "TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WR_UNCACHED_32B_sum"; do
    i=$((i+1))
    timeout 120 rocprofv3 --pmc $set -d $OUT/${build}_p$i -o p$i --output-format csv -- python3 $ROOT/tools/probe_local_corr_one.py $C $HS $G $R 64 8 > $OUT/${build}_p$i.log 2>&1
  done
done
cd $ROOT && python3 - <<PY
import csv, glob, collections, json, re
res = {}
for build in ("nt", "plain"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s_p*/**/*counter_collection.csv" % build, recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"local_corr_tile2_kernel<[^>]*>", r["Kernel_Name"])
            if m: agg[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res[build] = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
C, HS, G, R, B = $C, $HS, $G, $R, 64
planes = 4 * B * (2 * R + 1) ** 2 * G * G
out = {"source": "tools/pmc_write_local_corr.sh %d %d %d %d (rocprofv3 --pmc, one pass per counter set, 8 dispatches each, 64 directions)" % (C, HS, G, R),
       "output_plane_bytes_per_launch": planes, "builds": res}
json.dump(out, open("gpurun_out/local_corr_write_pmc_r%d.json" % R, "w"), indent=1)
print(json.dumps(out, indent=1))
PY
