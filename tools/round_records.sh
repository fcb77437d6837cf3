#!/bin/bash
# usage (GPU box): tools/round_records.sh <tag>  -- every measurement DESIGN.md quotes, from ONE tree on ONE box, into gpurun_out/rec_<tag>/
# (copy what is to be judged into profiles/ afterwards).  Each step under its own time limit; stops at the first failure.
set -o pipefail
T=${1:-r}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/rec_$T
mkdir -p $O
cd $R
step() { echo "== $1"; }
step "bench.py (default command)"
timeout -k 10 400 python bench.py > $O/bench_line.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cat $O/bench_line.json
step "rocprofv3 kernel stats of bench.py"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_c4 -o r -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_c4_line.json 2> $O/prof_c4.err ) || { tail -5 $O/prof_c4.err; exit 1; }
python3 tools/rocpd_summary.py $O/prof_c4/r_results.db $O/kernel_stats_c4.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline" && head -8 $O/kernel_stats_c4.csv | cut -c1-120
step "C1 kernel stats (tools/exp_c1.py)"
for dt in bf16 fp32; do
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof_c1_$dt -o r -- python3 $R/tools/exp_c1.py $dt > $O/c1_$dt.txt 2> $O/prof_c1_$dt.err ) || { tail -5 $O/prof_c1_$dt.err; exit 1; }
  python3 tools/rocpd_summary.py $O/prof_c1_$dt/r_results.db $O/c1_kernel_stats_$dt.csv "C1 10000 x 50000 x 128 $dt: rocprofv3 --kernel-trace --stats -- python3 tools/exp_c1.py $dt (23 knn_predict calls)" && cat $O/c1_$dt.txt
done
step "configs C1-C3"
timeout -k 10 400 python tools/bench_configs.py c1 c2 c3 > $O/configs.jsonl 2> $O/configs.err || { tail -5 $O/configs.err; exit 1; }
cut -c1-260 $O/configs.jsonl
step "HBM-bound kernels"
timeout -k 10 300 python tools/bench_hbm.py > $O/hbm_kernels.jsonl 2> $O/hbm.err || { tail -5 $O/hbm.err; exit 1; }
cut -c1-160 $O/hbm_kernels.jsonl
step "per-rank cost of the sharded job (steady state)"
bash tools/shard_cost.sh --steps 10 --warmup 3 || exit 1
cp gpurun_out/shard_cost.jsonl $O/shard_cost.jsonl
step "PMC traffic"
bash tools/pmc_traffic.sh $T | tail -1 > $O/traffic.json || exit 1
cat $O/traffic.json
step "PMC SQ counters"
bash tools/pmc_round.sh $T > $O/pmc_sim_kernel.csv 2> $O/pmc.err || { tail -5 $O/pmc.err; exit 1; }
head -40 $O/pmc_sim_kernel.csv
echo "== done"
