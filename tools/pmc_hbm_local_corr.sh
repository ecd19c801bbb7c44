#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_hbm_local_corr.sh C HS G R [OUTJSON] -> FETCH_SIZE / WRITE_SIZE per dispatch of the
# local-correlation kernels (separate --pmc passes, MI355X_MICROARCH.md HBM section), written as profiles-style JSON
C=$1; HS=$2; G=$3; R=$4; JSON=${5:-gpurun_out/local_corr_pmc.json}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/hbm_lc_r$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set -d $OUT/p$i -o p$i --output-format csv -- python3 $ROOT/tools/probe_local_corr_one.py $C $HS $G $R 64 8 > $OUT/p$i.log 2>&1
done
cd $ROOT && python3 - <<PY
import csv, glob, collections, json, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"local_corr_\w+<[^>]*>", r["Kernel_Name"])
        if m: agg[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
C, HS, G, R, B = $C, $HS, $G, $R, 64
algo = 4 * B * (C * G * G + C * HS * HS + 2 * G * G + (2 * R + 1) ** 2 * G * G)
out = {"source": "rocprofv3 --pmc FETCH_SIZE and rocprofv3 --pmc WRITE_SIZE (separate passes), tools/pmc_hbm_local_corr.sh %d %d %d %d: "
                 "tools/probe_local_corr_one.py, 64 directions, homography flows, 8 dispatches each" % (C, HS, G, R),
       "correction": "gfx950 FETCH_SIZE counts 64 B per 128-B request: doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE taken as is; both in KB",
       "algorithmic_bytes_per_launch": algo, "kernels": {}}
tot = 0.0
for k, d in agg.items():
    if "local_corr" not in k: continue
    f = sum(d.get("FETCH_SIZE", [0])) / max(len(d.get("FETCH_SIZE", [1])), 1)
    w = sum(d.get("WRITE_SIZE", [0])) / max(len(d.get("WRITE_SIZE", [1])), 1)
    b = (2 * f + w) * 1024
    out["kernels"][k] = {"FETCH_SIZE_KB_per_launch": round(f, 1), "WRITE_SIZE_KB_per_launch": round(w, 1), "hbm_bytes_per_launch": int(b)}
    tot += b
out["hbm_bytes_per_launch"] = int(tot)
out["note"] = "sum over the launches of one gfn_local_corr_fwd call (plan + tile kernel [+ second launch]): %.1f MB vs %.1f MB algorithmic" % (tot / 1e6, algo / 1e6)
json.dump(out, open("$JSON", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
