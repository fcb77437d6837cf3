#!/bin/bash
# usage: tools/build_variant.sh <suffix> <extra hipcc flags...>   -> image-classification_amd/libmi_knn_<suffix>.so (experiments only)
S=$1; shift
cd "$(dirname "$0")/../image-classification_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC "$@" -o ../libmi_knn_$S.so mi_knn.hip
