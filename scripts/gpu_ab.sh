#!/bin/bash
# A/B of pre-built libraries on one box: bash scripts/gpu_ab.sh build/lib_x.so [build/lib_y.so ...]; each is timed with scripts/time_march.py
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in 1 2; do
for lib in base "$@"; do
  if [ "$lib" = base ]; then unset IFF_LIB_PATH; else export IFF_LIB_PATH="$PWD/$lib"; fi      # never copied over the product library
  echo "== $lib"
  timeout -k 10 300 python scripts/time_march.py ${CFG:-lego16k} 2>/dev/null
done
done
