#!/bin/bash
# Dev aid: the default cold bench (short) under each of the given pre-built libraries, same box, two rounds.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in 1 2; do
for lib in base "$@"; do
  if [ "$lib" = base ]; then unset IFF_LIB_PATH; else export IFF_LIB_PATH="$PWD/$lib"; fi      # never copied over the product library
  timeout -k 10 300 python bench.py --config ${CFG:-lego16k} --steps 150 --warmup 15 --no-cpu-baseline --no-instrument 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', j['value'], j['ms_per_step'])"
done
done
