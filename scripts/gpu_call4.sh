#!/bin/bash
# dev: ray-gradient checks
mkdir -p gpurun_out
python -m pytest tests/test_hip_dropin.py -m gpu -q -x -s -k "ray_gradients or pose_refinement" > gpurun_out/grad_tests.log 2>&1
tail -40 gpurun_out/grad_tests.log
