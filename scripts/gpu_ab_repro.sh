#!/bin/bash
# Dev aid: scripts/replay_vs_eager_stages.py (ONLY=trunk: the march next to the trunk launch) under each of the given pre-built
# libraries (build/lib_<tag>.so), same box.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for lib in ${BASE:-base} "$@"; do
  if [ "$lib" = base ]; then unset IFF_LIB_PATH; else export IFF_LIB_PATH="$PWD/$lib"; fi      # never copied over the product library
  echo "== $lib"
  ONLY=${ONLY-trunk} CONFIG=${CONFIG:-truck32k} ROUNDS=${ROUNDS:-200} timeout -k 10 700 python scripts/replay_vs_eager_stages.py > gpurun_out/repro_$(basename $lib .so).log 2>&1
  tail -1 gpurun_out/repro_$(basename $lib .so).log
done
