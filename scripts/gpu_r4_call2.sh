#!/bin/bash
# round 4, call 2: GPU suite on the corner-bit occupancy; the packed-fp32 fault: destination over the broadcast pair (7) vs never (8)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
echo "== pytest $(date +%T)"
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/c2_pytest.log 2>&1; rc=$?; tail -n 5 gpurun_out/c2_pytest.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
echo "== repro $(date +%T)"
BASE=build/lib_pklerp.so ROUNDS=${ROUNDS:-500} bash scripts/gpu_ab_repro.sh build/lib_lerp7.so build/lib_lerp8.so build/lib_nopk.so base || exit 1
echo "== bench A/B $(date +%T)"
bash scripts/gpu_ab_bench.sh build/lib_nopk.so build/lib_lerp8.so
echo "== done $(date +%T)"
