#!/bin/bash
# usage (GPU box, repo root): bash tools/round_profiles.sh rNN  -> bench JSON lines (with the CPU leg) and rocprofv3 kernel stats of every workload
R=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
run() {  # name, bench args...
  NAME=$1; shift
  python3 bench.py --steps 20 --warmup 3 "$@" > gpurun_out/${R}_bench_$NAME.json 2> gpurun_out/${R}_bench_$NAME.err
  bash tools/prof_bench.sh ${R}_prof_$NAME "$@" > /dev/null 2>&1
  python3 -c "
import json,sys
j=json.load(open('gpurun_out/${R}_bench_$NAME.json'))
print('$NAME', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['avg_launch_us'], j.get('cpu_baseline',{}).get('value'), j.get('mean_corner_error_vs_ref_px'))"
}
run 448b32
run 672b16 --workload 672b16
run pyr_fp16 --workload pyr-fp16
run 448b32_convstack_amp --conv-stack amp
run 448b32_convstack_fp16 --conv-stack fp16
run 448b32_convstack_fp32 --conv-stack fp32
