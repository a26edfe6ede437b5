#!/bin/bash
# Dev aid: scripts/replay_vs_eager_stages.py (ONLY=trunk: the march next to the trunk launch) under each of the given pre-built
# libraries (build/lib_<tag>.so), same box.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
cp iffnerf_amd/libiffnerf_hip.so /tmp/lib_keep.so
for lib in ${BASE:-base} "$@"; do
  if [ "$lib" = base ]; then cp /tmp/lib_keep.so iffnerf_amd/libiffnerf_hip.so; else cp "$lib" iffnerf_amd/libiffnerf_hip.so; fi
  echo "== $lib"
  ONLY=${ONLY-trunk} CONFIG=${CONFIG:-truck32k} ROUNDS=${ROUNDS:-200} timeout -k 10 700 python scripts/replay_vs_eager_stages.py 2>&1 | tail -1
done
cp /tmp/lib_keep.so iffnerf_amd/libiffnerf_hip.so
