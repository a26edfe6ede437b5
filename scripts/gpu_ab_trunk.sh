#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
cp iffnerf_amd/libiffnerf_hip.so /tmp/lib_keep.so
for lib in base "$@"; do
  if [ "$lib" = base ]; then cp /tmp/lib_keep.so iffnerf_amd/libiffnerf_hip.so; else cp "$lib" iffnerf_amd/libiffnerf_hip.so; fi
  echo "== $lib"; timeout -k 10 300 python scripts/ab_trunk.py 2>/dev/null | tail -2
done
cp /tmp/lib_keep.so iffnerf_amd/libiffnerf_hip.so
