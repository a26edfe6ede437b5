#!/bin/bash
# Dev aid: scripts/time_vit32.py (32 images, one stream) under the in-tree library and under each build/lib_<tag>.so named.
#     FORMS=2,4 PRECS=fp32 bash scripts/gpu_vit_ab.sh noremap abl1 ...
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in 1 2; do
for lib in base "$@"; do
  if [ "$lib" = base ]; then unset IFF_LIB_PATH; else export IFF_LIB_PATH="$PWD/build/lib_$lib.so"; fi
  ONLY32=1 timeout -k 10 200 python scripts/time_vit32.py 2>/dev/null | sed "s/^/$lib /" | cut -c1-200
done
done
