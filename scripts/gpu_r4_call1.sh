#!/bin/bash
# round 4, call 1: GPU suite on the boundary change; the packed-fp32 fault under hand-placed instruction forms; nopk library A/B
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
echo "== pytest $(date +%T)"
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/c1_pytest.log 2>&1; rc=$?; tail -n 5 gpurun_out/c1_pytest.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
echo "== repro $(date +%T)"
BASE=build/lib_pklerp.so ROUNDS=${ROUNDS:-120} bash scripts/gpu_ab_repro.sh build/lib_lerp1.so build/lib_lerp2.so build/lib_lerp3.so build/lib_lerp4.so build/lib_lerp6.so build/lib_nopk.so || exit 1
echo "== bench A/B $(date +%T)"
bash scripts/gpu_ab_bench.sh build/lib_nopk.so
echo "== done $(date +%T)"
