#!/bin/bash
# dev: in-flight / batch sweep of the cold path
mkdir -p gpurun_out
: > gpurun_out/sweep.log
for nf in 3 4 6; do
  for b in 8 16 24; do
    python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-instrument --in-flight $nf --batch $b 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('in_flight $nf batch $b', d['value'], d['ms_per_step'])" >> gpurun_out/sweep.log || exit 1
  done
done
cat gpurun_out/sweep.log
