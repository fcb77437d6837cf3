#!/bin/bash
# usage (GPU box): tools/ab_c1.sh "<lib-suffix> ..." [rounds] -- interleaved A/B of library variants at C1 (tools/exp_c1.py, both dtypes) on ONE box
R=${2:-2}
for r in $(seq 1 $R); do for s in $1; do lib=$GRAFT_REPO_ROOT/image-classification_amd/libmi_knn$s.so
  [ "$s" = "base" ] && lib=$GRAFT_REPO_ROOT/image-classification_amd/libmi_knn.so
  for dt in bf16 fp32; do MI_KNN_LIB=$lib timeout -k 10 120 python tools/exp_c1.py $dt 2>&1 | grep "similarity kernels" || exit 1; done
done; done
