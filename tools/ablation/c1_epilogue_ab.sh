#!/bin/bash
# usage (GPU box): tools/ablation/c1_epilogue_ab.sh [fp32|bf16] -- per-launch kernel times at C1 for the timing-only epilogue variants
# (make_epilogue_variants.py, make_tauinf_variant.py).  Only the FIRST filter launch is comparable across variants (same thresholds).
DT=${1:-bf16}
for v in "" _tauinf _nohit _nostore _noatomic _nomask _nocnt; do
  export MI_KNN_LIB=$GRAFT_REPO_ROOT/image-classification_amd/libmi_knn$v.so
  echo "== variant '${v}'"
  bash $GRAFT_REPO_ROOT/tools/trace_c1.sh ab$v $DT | grep -E "sim_kernel|knn_predict|refresh" | head -9 || exit 1
done
