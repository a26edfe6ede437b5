#!/bin/bash
# dev: sharded path at world size 1 (RCCL group of one) vs the single-GPU path
mkdir -p gpurun_out
for extra in "" "--force-sharded"; do
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-instrument $extra > gpurun_out/k4b_bench.log 2>&1 || { tail -20 gpurun_out/k4b_bench.log; exit 1; }
tail -1 gpurun_out/k4b_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$extra', d['value'], d['ms_per_step'], d['config']['launch'][:60])"
done
