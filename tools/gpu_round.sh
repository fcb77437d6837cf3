#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_round.sh <tag>
# GPU tests, default bench line, then a rocprofv3 kernel-stats pass of a short bench; every step under its own time limit.
set -o pipefail
T=${1:-x}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $R/gpurun_out/t_$T.log 2>&1 || { tail -30 $R/gpurun_out/t_$T.log; exit 1; }
tail -3 $R/gpurun_out/t_$T.log
timeout -k 10 400 python bench.py > $R/gpurun_out/b_$T.json 2> $R/gpurun_out/b_$T.err || { tail -20 $R/gpurun_out/b_$T.err; exit 1; }
cat $R/gpurun_out/b_$T.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$T -o r -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_$T.json 2> $R/gpurun_out/prof_$T.err || { tail -20 $R/gpurun_out/prof_$T.err; exit 1; }
find $R/gpurun_out/prof_$T -name '*kernel_stats.csv' | head -1
