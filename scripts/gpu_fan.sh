#!/bin/bash
# fan kernel: parity test vs the general kernels, then the march timed under the three IFF_MARCH_FAN settings
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
IFF_MARCH_FAN=2 timeout -k 10 600 python -m pytest tests/test_hip_field.py -m gpu -q -x > gpurun_out/fan_test.log 2>&1; rc=$?
tail -n 25 gpurun_out/fan_test.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
if [ $rc -ne 0 ]; then exit $rc; fi
for m in ${FAN_MODES:-0 2}; do
  IFF_MARCH_FAN=$m timeout -k 10 300 python scripts/time_march.py ${1:-lego16k} 2> gpurun_out/time_march_$m.err | tee gpurun_out/time_march_$m.json
  rc=${PIPESTATUS[0]}; if [ $rc -ne 0 ]; then tail -5 gpurun_out/time_march_$m.err; exit $rc; fi
done
