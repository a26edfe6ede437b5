#!/bin/bash
# dev: identify/sharded tests + two bench repeats
mkdir -p gpurun_out
python -m pytest tests/test_hip_identify.py tests/test_hip_fullsize.py tests/test_hip_sharded.py tests/test_hip_dropin.py -m gpu -q -x > gpurun_out/all_tests.log 2>&1 || { tail -40 gpurun_out/all_tests.log; exit 1; }
tail -2 gpurun_out/all_tests.log
for i in 1 2; do
python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/k4b_bench.log 2>&1
tail -1 gpurun_out/k4b_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms'], {k[:4]:v.get('avg_launch_ms') for k,v in d['roofline']['other_kernels'].items()}, d['roofline']['avg_launch_ms'], d['warm_poses_per_s'])"
done
