#!/bin/bash
# usage (GPU box, repo root): bash tools/round6_profiles.sh  -> everything profiles/r06_* of the final tree is made from (bench lines, rocprofv3 kernel
# stats, per-call roofline table, HBM counters of the roofline kernel); results under gpurun_out/, copied into profiles/ by hand
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/round_profiles.sh r06 > gpurun_out/r06_round_profiles.log 2>&1
python3 bench.py > gpurun_out/r06_bench_default_line.json 2> gpurun_out/r06_bench_default_line.err
python3 tools/local_corr_roofline.py > gpurun_out/r06_local_corr_roofline.md 2> /dev/null
python3 tools/local_corr_roofline.py --workload 672b16 >> gpurun_out/r06_local_corr_roofline.md 2> /dev/null
bash tools/pmc_hbm_local_corr.sh 32 112 64 4 gpurun_out/r06_local_corr_pmc.json > /dev/null 2>&1
bash tools/pmc_local_corr_r6.sh 32 112 64 4 r06 > /dev/null 2>&1
cp gpurun_out/pmc6_r4_r06/summary.txt gpurun_out/r06_local_corr_sq_counters.txt
cat gpurun_out/r06_round_profiles.log
