#!/bin/bash
# usage (GPU box): tools/prof_configs.sh <tag> <c1|c2|c3> -- rocprofv3 kernel stats of one tools/bench_configs.py case
set -o pipefail
T=$1; C=$2; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$T -o r -- python3 $R/tools/bench_configs.py $C > $R/gpurun_out/prof_$T.jsonl 2> $R/gpurun_out/prof_$T.err || { tail -5 $R/gpurun_out/prof_$T.err; exit 1; }
cd $R && python3 tools/rocpd_summary.py gpurun_out/prof_$T/r_results.db gpurun_out/$T.csv "$T $C" && head -24 gpurun_out/$T.csv | cut -c1-130
