#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
for lib in base "$@"; do
  if [ "$lib" = base ]; then unset IFF_LIB_PATH; else export IFF_LIB_PATH="$PWD/$lib"; fi      # never copied over the product library
  echo "== $lib"; timeout -k 10 300 python scripts/ab_trunk.py 2>/dev/null | tail -2
done
