#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_image_side.py tests/test_table_files.py -m gpu -q --durations=5 > gpurun_out/pytest_img.log 2>&1
rc=$?; echo "rc=$rc"; tail -40 gpurun_out/pytest_img.log
