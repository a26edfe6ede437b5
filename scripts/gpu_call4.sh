#!/bin/bash
# dev: batch / in-flight sweep of the larger configs
mkdir -p gpurun_out
: > gpurun_out/sweep.log
for cfg in bicycle64k truck32k; do
  for b in 4 8 16; do
    for nf in 2 4; do
    python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline --no-instrument --in-flight $nf --batch $b 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$cfg batch $b in_flight asked $nf got', d['config']['steps_in_flight'], d['value'], d['ms_per_step'])" >> gpurun_out/sweep.log || exit 1
    done
  done
done
cat gpurun_out/sweep.log
