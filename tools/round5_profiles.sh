#!/bin/bash
# usage (GPU box, repo root): bash tools/round5_profiles.sh  -> everything profiles/r05_* is made from (bench lines, rocprofv3 kernel stats,
# per-call roofline table, HBM and SQ counters of the roofline kernel); results under gpurun_out/, copied into profiles/ by hand
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/round_profiles.sh r05 > gpurun_out/r05_round_profiles.log 2>&1
python3 bench.py > gpurun_out/r05_bench_default_line.json 2> gpurun_out/r05_bench_default_line.err
python3 bench.py --steps 20 --warmup 3 --graphs > gpurun_out/r05_bench_448b32_graphs.json 2> gpurun_out/r05_bench_448b32_graphs.err
python3 tools/local_corr_roofline.py > gpurun_out/r05_local_corr_roofline.md 2> /dev/null
python3 tools/local_corr_roofline.py --workload 672b16 >> gpurun_out/r05_local_corr_roofline.md 2> /dev/null
bash tools/pmc_hbm_local_corr.sh 32 112 64 4 gpurun_out/r05_local_corr_pmc.json > /dev/null 2>&1
bash tools/pmc_local_corr_r5.sh 32 112 64 4 r05 > /dev/null 2>&1
cp gpurun_out/pmc5_r4_r05/summary.txt gpurun_out/r05_local_corr_sq_counters.txt
python3 tools/ablate_lean.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_ablate_lean.txt
cat gpurun_out/r05_round_profiles.log
