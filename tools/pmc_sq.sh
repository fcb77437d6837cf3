#!/bin/bash
# usage (GPU box): tools/pmc_sq.sh "<lib suffixes>"  -- one SQ counter pass per library variant over a one-chunk bench
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for S in $1; do
  LIB=$R/image-classification_amd/libmi_knn$S.so; [ "$S" = "base" ] && LIB=$R/image-classification_amd/libmi_knn.so
  mkdir -p $R/gpurun_out/pmcsq/$S
  MI_KNN_LIB=$LIB timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmcsq/$S -o r -- python3 $R/bench.py --steps 1 --warmup 0 --queries 16384 --query-chunk 16384 --no-cpu-baseline > $R/gpurun_out/pmcsq/$S.json 2> $R/gpurun_out/pmcsq/$S.log || { tail -5 $R/gpurun_out/pmcsq/$S.log; exit 1; }
  echo "== $S"; python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcsq/$S sim_kernel
done
