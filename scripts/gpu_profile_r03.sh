#!/bin/bash
# Round-3 evidence run on the GPU box: for every bench workload the rocprofv3 kernel statistics and the PMC passes
# (scripts/profile_pmc.sh), summarised per config into gpurun_out/r03_* (copy what should be judged into profiles/).
#     bash scripts/gpu_profile_r03.sh [config ...]        default: lego16k truck32k bicycle64k
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
CFGS=("$@"); [ ${#CFGS[@]} -eq 0 ] && CFGS=(lego16k truck32k bicycle64k)
for cfg in "${CFGS[@]}"; do
  echo "=== $cfg $(date +%T)"
  bash scripts/profile_pmc.sh r03_$cfg --config $cfg > gpurun_out/profile_r03_$cfg.log 2>&1 || { tail -20 gpurun_out/profile_r03_$cfg.log; exit 1; }
  P=gpurun_out/prof_r03_$cfg
  python3 scripts/summarize_pmc.py gpurun_out/r03_$cfg $cfg $P/sq1 $P/sq2 $P/sq3 $P/tcc $P/tcp $P/fetch $P/write $P/grbm > gpurun_out/r03_${cfg}_summary.txt 2>&1
  mv gpurun_out/r03_${cfg}_hbm_traffic.json gpurun_out/r03_hbm_traffic_$cfg.json
  mv gpurun_out/r03_${cfg}_pmc_counters.csv gpurun_out/r03_pmc_counters_$cfg.csv
  cp "$(ls $P/stats/*/*kernel_stats.csv $P/stats/*kernel_stats.csv 2>/dev/null | head -1)" gpurun_out/r03_bench_kernel_stats_$cfg.csv
  cp "$(ls $P/stats_if1/*/*kernel_stats.csv $P/stats_if1/*kernel_stats.csv 2>/dev/null | head -1)" gpurun_out/r03_bench_kernel_stats_inflight1_$cfg.csv
  tail -12 gpurun_out/r03_${cfg}_summary.txt | cut -c1-400
  rm -rf "$P"        # the raw traces (20 MB per config): gpurun copies back at most 64 MiB
done
ls -la gpurun_out/r03_*
