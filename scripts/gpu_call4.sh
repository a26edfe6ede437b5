#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_dropin.py -m gpu -q -x > gpurun_out/pytest_dropin.log 2>&1
rc=$?; echo "rc=$rc"; tail -30 gpurun_out/pytest_dropin.log
