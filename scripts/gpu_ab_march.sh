#!/bin/bash
# Dev aid: GPU suite on the in-tree library, then the march alone and the default bench under each given development build, same box.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_field.py tests/test_hip_fullsize.py tests/test_hip_dropin.py -m gpu -q -x > gpurun_out/ab_pytest.log 2>&1; rc=$?; tail -n 4 gpurun_out/ab_pytest.log
if [ $rc -ne 0 ]; then exit 1; fi
bash scripts/gpu_ab.sh "$@"
bash scripts/gpu_ab_bench.sh "$@"
