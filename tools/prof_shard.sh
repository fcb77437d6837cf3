#!/bin/bash
# usage (GPU box): tools/prof_shard.sh <tag> <bank rows> -- rocprofv3 kernel stats of bench.py on one emulated shard
set -o pipefail
T=$1; ROWS=$2; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$T -o r -- python3 $R/bench.py --bank $ROWS --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_$T.json 2> $R/gpurun_out/prof_$T.err || { tail -5 $R/gpurun_out/prof_$T.err; exit 1; }
cd $R && python3 tools/rocpd_summary.py gpurun_out/prof_$T/r_results.db gpurun_out/$T.csv "$T rows=$ROWS" && grep "mi::" gpurun_out/$T.csv | cut -c1-120
