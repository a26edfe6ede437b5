#!/bin/bash
# same-box A/B of the fused fan kernels on the march alone: bash scripts/gpu_ab_march.sh [config ...]   (run on the GPU box)
# FAN_WAVES=4: k4f_fan_march (four waves per fan, register-staged patches); 8: k4g_fan_march (eight waves, DMA-staged); 0: the handle's choice
set -e
for cfg in "${@:-lego16k}"; do
  for rep in 1 2; do
    for w in 4 8 0; do
      FAN_WAVES=$w python scripts/time_march.py $cfg 2>/dev/null | tail -1
    done
  done
done
