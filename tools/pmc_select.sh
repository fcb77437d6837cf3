#!/bin/bash
# usage (GPU box): tools/pmc_select.sh -- SQ counters of the refresh kernels on an emulated 1/8 shard (one pass)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmcsel
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmcsel -o r -- python3 $R/bench.py --steps 1 --warmup 0 --queries 16384 --bank 160000 --no-cpu-baseline > $R/gpurun_out/pmcsel/run.json 2> $R/gpurun_out/pmcsel/run.log || { tail -5 $R/gpurun_out/pmcsel/run.log; exit 1; }
for k in "wave_select_kernel<0>" "wave_select_kernel<1>" merge_kernel; do echo "== $k"; python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcsel "$k"; done
