#!/bin/bash
# round 4, call 5: GPU suite (4-lane head in the fused march, merge test), bench, VALU counters of the march
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
echo "== pytest $(date +%T)"
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/c5_pytest.log 2>&1; rc=$?; tail -n 8 gpurun_out/c5_pytest.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
echo "== bench $(date +%T)"
for i in 1 2; do timeout -k 10 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/c5_bench$i.json 2> gpurun_out/c5_bench.err; rc=$?; python - <<PY
import json
d=json.loads(open("gpurun_out/c5_bench$i.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d.get("stage_ms"))
PY
done
echo "== pmc march $(date +%T)"
bash scripts/pmc_march.sh r4c5 lego16k 2>&1 | grep -E "sq1|stats" | cut -c1-400
echo "== done $(date +%T)"
