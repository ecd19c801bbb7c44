#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_bench.sh NAME [bench args...]  -> rocprofv3 kernel stats of bench.py into gpurun_out/NAME
NAME=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/$NAME -o kt --output-format csv -- python3 $ROOT/bench.py --steps 10 --warmup 2 --cpu-pairs 0 --no-stack-leg --no-other-workloads --no-stress-legs "$@" > $ROOT/gpurun_out/$NAME.json 2> $ROOT/gpurun_out/$NAME.err
cd $ROOT && python3 tools/summarize_rocprof.py gpurun_out/$NAME $NAME "bench.py --steps 10 --warmup 2 --cpu-pairs 0 --no-stack-leg --no-other-workloads --no-stress-legs $* (27 steps: 2 warm-up, 10 timed (two streams), 1 + 10 on one stream with the roofline op's events, the three plan-attribution steps and the counter step)" > /dev/null
mv profiles/$NAME.md profiles/${NAME}_kernel_stats.csv gpurun_out/ 2>/dev/null
head -32 gpurun_out/$NAME.md
