#!/bin/bash
# dev: three bench repeats
mkdir -p gpurun_out
for i in 1 2 3; do
python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/k4b_bench.log 2>&1
tail -1 gpurun_out/k4b_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms'], {k[:4]:v.get('avg_launch_ms') for k,v in d['roofline']['other_kernels'].items()}, d['roofline']['avg_launch_ms'])"
done
