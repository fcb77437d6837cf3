#!/bin/bash
# usage (GPU box): tools/pmc_round.sh <tag> [lib suffix]  -- one rocprofv3 --pmc pass per counter group over a one-chunk bench
set -o pipefail
T=$1; S=$2
R=$GRAFT_REPO_ROOT
LIB=$R/image-classification_amd/libmi_knn$S.so
mkdir -p $R/gpurun_out/pmc_$T
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES" \
           "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  MI_KNN_LIB=$LIB timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_$T/p$i -o r -- python3 $R/bench.py --steps 1 --warmup 0 --queries 16384 --query-chunk 16384 --no-cpu-baseline > $R/gpurun_out/pmc_$T/p$i.json 2> $R/gpurun_out/pmc_$T/p$i.log || { tail -5 $R/gpurun_out/pmc_$T/p$i.log; exit 1; }
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_$T sim_kernel
