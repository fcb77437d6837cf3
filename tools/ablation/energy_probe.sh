#!/bin/bash
# usage (GPU box): tools/ablation/energy_probe.sh [rounds] -- interleaved rounds of toggle_probe.py (random x random, one C4-sized
# chunk, in-kernel clock) over the make_energy_variants.py libraries on ONE box -> gpurun_out/energy_probe.txt
R=${1:-2}
: > gpurun_out/energy_probe.txt
for r in $(seq 1 $R); do for v in ${VARIANTS:-e_base e_halfreads e_l2all e_morereads}; do
  echo "== $v (round $r)" | tee -a gpurun_out/energy_probe.txt
  MI_KNN_LIB=$GRAFT_REPO_ROOT/image-classification_amd/libmi_knn_$v.so PROBE_RANDOM_ONLY=1 timeout -k 10 200 python tools/ablation/toggle_probe.py 2>&1 | tee -a gpurun_out/energy_probe.txt || exit 1
done; done
