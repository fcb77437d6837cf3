#!/bin/bash
# usage (GPU box): tools/pmc_c1.sh <tag> [bf16|fp32] -- rocprofv3 --pmc passes (one counter group per run) over tools/exp_c1.py (C1: 10k x 50k x 128);
# per-dispatch means for the filter launch of the configuration (sim_walk_kernel at bf16, sim_kernel at fp32) and for bag_refresh_kernel
set -o pipefail
T=$1; DT=${2:-bf16}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_$T
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_FLAT"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_$T/p$i -o r -- python3 $R/tools/exp_c1.py $DT > $R/gpurun_out/pmc_$T/p$i.txt 2> $R/gpurun_out/pmc_$T/p$i.log || { tail -5 $R/gpurun_out/pmc_$T/p$i.log; exit 1; }
done
K=sim_walk_kernel; [ "$DT" = "fp32" ] && K="sim_kernel<0, 1"
echo "## $K"; python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_$T "$K"
echo "## bag_refresh_kernel"; python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_$T bag_refresh_kernel
