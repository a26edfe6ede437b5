#!/bin/bash
# every bench.py workload once (short runs), plus the sharded code path at world size 1
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() {
    local tag=$1; shift
    echo "== $tag $(date +%T)"
    timeout -k 10 300 python bench.py "$@" > gpurun_out/cfg_$tag.json 2> gpurun_out/cfg_$tag.err
    rc=$?; echo "   rc=$rc"
    if [ $rc -ne 0 ]; then tail -15 gpurun_out/cfg_$tag.err; fi
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
    python - "$tag" <<'PY'
import json,sys
try:
    j=json.loads(open(f'gpurun_out/cfg_{sys.argv[1]}.json').read().strip().splitlines()[-1])
    r=j.get('roofline') or {}
    print('  ', j['value'], 'poses/s', j['ms_per_step'], 'ms/step', 'in-flight', j['config']['steps_in_flight'], 'q/step', j['config']['queries_per_step'], 'trunk ms', r.get('avg_launch_ms'), 'frac', r.get('frac'), 'cpu', (j.get('cpu_baseline') or {}).get('value'))
except Exception as e:
    print('  parse failed', e)
PY
}
run lego16k --steps 100 --warmup 10 --no-cpu-baseline
run truck32k --config truck32k --steps 40 --warmup 5 --no-cpu-baseline
run bicycle64k --config bicycle64k --steps 30 --warmup 5 --no-cpu-baseline
run lego_b64 --config lego_b64 --steps 60 --warmup 10 --no-cpu-baseline
run lego540k --config lego540k --steps 12 --warmup 3 --no-cpu-baseline
run lego16k_sharded_ws1 --steps 100 --warmup 10 --no-cpu-baseline --no-instrument --force-sharded
run lego_b64_sharded_ws1 --config lego_b64 --steps 60 --warmup 10 --no-cpu-baseline --no-instrument --force-sharded
