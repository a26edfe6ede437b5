#!/bin/bash
# round 4, call 3: GPU suite (ViT fp32 class, corner-bit occupancy, stage-level concurrency test), bench, VALU counters of the march
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
echo "== pytest $(date +%T)"
timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/c3_pytest.log 2>&1; rc=$?; tail -n 25 gpurun_out/c3_pytest.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
echo "== bench $(date +%T)"
timeout -k 10 600 python bench.py --steps 150 --warmup 15 > gpurun_out/c3_bench.json 2> gpurun_out/c3_bench.err; rc=$?; echo "rc=$rc"; tail -c 1500 gpurun_out/c3_bench.json; tail -3 gpurun_out/c3_bench.err
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
echo "== pmc march $(date +%T)"
bash scripts/pmc_march.sh r4c3 lego16k 2>&1 | tail -12
echo "== done $(date +%T)"
