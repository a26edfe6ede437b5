#!/bin/bash
# fan kernel: parity test, march timing (general vs fused), then the per-phase timeline from a stamps build kept in gpurun_out/
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
IFF_MARCH_FAN=2 timeout -k 10 600 python -m pytest tests/test_hip_field.py -m gpu -q -x > gpurun_out/fan_test.log 2>&1; rc=$?
tail -n 5 gpurun_out/fan_test.log
if [ $rc -ne 0 ]; then tail -n 40 gpurun_out/fan_test.log; exit $rc; fi
for m in ${FAN_MODES:-0 2}; do
  IFF_MARCH_FAN=$m timeout -k 10 300 python scripts/time_march.py ${1:-lego16k} 2> gpurun_out/time_march_$m.err | tee gpurun_out/time_march_$m.json
  rc=${PIPESTATUS[0]}; if [ $rc -ne 0 ]; then tail -5 gpurun_out/time_march_$m.err; exit $rc; fi
done
if [ -f build/lib_stamps.so ]; then
  cp iffnerf_amd/libiffnerf_hip.so /tmp/lib_keep.so && cp build/lib_stamps.so iffnerf_amd/libiffnerf_hip.so
  IFF_MARCH_FAN=2 timeout -k 10 300 python scripts/fan_stamps.py 2>/dev/null | tail -17
  cp /tmp/lib_keep.so iffnerf_amd/libiffnerf_hip.so
fi
