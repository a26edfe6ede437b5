#!/bin/bash
# The packed-fp32 investigation (csrc/fan_march_kernels.hip lerp_plane_q), as it was run in round 4.
#   bash scripts/packed_fp32_forms.sh build      here (CPU): the libraries -- the compiler's own packing of the tap combination
#                                                (+packed-fp32-ops: the form that fails) and the hand-placed instruction forms
#   gpurun -- 'bash scripts/packed_fp32_forms.sh run'     on the GPU box: the stage-level replay-against-eager check under each
# Result of the round-4 runs (2 000 checked steps each, truck32k, four graphs in flight): compiler packing 14 mismatches (rays 6, 7 /
# 14, 15 / 22, 23 of one of the last ~300 tiles each); every hand-placed form 0; no packed fp32 at all (the product build) 0.
# The object-file and basic-block hybrids that separated victim (march kernel) and aggressor (trunk kernel): scripts/asm_variant.py.
set -u
cd "$(dirname "$0")/.."
PK="-Xclang -target-feature -Xclang +packed-fp32-ops"
if [ "${1:-}" = build ]; then
  python -m iffnerf_amd.build --tag pklerp -- $PK | tail -1
  for v in 1 2 3 4 6 7 8; do python -m iffnerf_amd.build --tag lerp$v -- $PK -DFAN_LERP_ASM=$v | tail -1; done
  exit 0
fi
mkdir -p gpurun_out
BASE=build/lib_pklerp.so ROUNDS=${ROUNDS:-500} bash scripts/gpu_ab_repro.sh build/lib_lerp1.so build/lib_lerp2.so build/lib_lerp3.so build/lib_lerp4.so \
  build/lib_lerp6.so build/lib_lerp7.so build/lib_lerp8.so base
