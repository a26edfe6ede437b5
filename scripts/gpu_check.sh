#!/bin/bash
# One GPU-box call: the GPU test suite, then a short default bench.  Usage: bash scripts/gpu_check.sh [pytest args...]
# A step that times out or is killed ends the call (no further GPU step after a hang).
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
step() {   # name, timeout, command...
    local name=$1 t=$2; shift 2
    echo "== $name $(date +%T)"
    timeout -k 10 "$t" "$@" > "gpurun_out/$name.log" 2> "gpurun_out/$name.err"
    local rc=$?
    echo "   rc=$rc"; tail -n 12 "gpurun_out/$name.log" | cut -c1-1500
    if [ $rc -ne 0 ]; then tail -n 30 "gpurun_out/$name.err"; fi
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$name timed out: stopping"; exit $rc; fi
    return 0
}
step pytest 1000 python -m pytest tests -m gpu -q --durations=8 "$@"
step bench 400 python bench.py --steps 150 --warmup 15
