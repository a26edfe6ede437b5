#!/bin/bash
# Dev aid: core clock / power of the GPU while bench.py runs (rocm-smi polled once a second).  bash scripts/gpu_clocks.sh [bench args]
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 400 python bench.py --steps 6000 --warmup 20 --no-cpu-baseline --no-instrument "$@" > gpurun_out/clk_bench.json 2>/dev/null &
bp=$!
: > gpurun_out/clocks.log
for i in $(seq 1 200); do
  if ! kill -0 $bp 2>/dev/null; then break; fi
  echo "t=$i" >> gpurun_out/clocks.log
  rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|Power|GPU use|fclk|mclk" >> gpurun_out/clocks.log
  sleep 1
done
wait $bp
tail -c 400 gpurun_out/clk_bench.json | head -c 400
