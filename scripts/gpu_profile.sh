#!/bin/bash
# Evidence run on the GPU box: for every bench workload the rocprofv3 kernel statistics and the PMC passes
# (scripts/profile_pmc.sh), summarised per config into gpurun_out/<tag>_* (copy what should be judged into profiles/).
#     TAG=r04 bash scripts/gpu_profile.sh [config ...]        default: lego16k truck32k bicycle64k
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
T=${TAG:-r04}
CFGS=("$@"); [ ${#CFGS[@]} -eq 0 ] && CFGS=(lego16k truck32k bicycle64k)
for cfg in "${CFGS[@]}"; do
  echo "=== $cfg $(date +%T)"
  bash scripts/profile_pmc.sh ${T}_$cfg --config $cfg > gpurun_out/profile_${T}_$cfg.log 2>&1 || { tail -20 gpurun_out/profile_${T}_$cfg.log; exit 1; }
  P=gpurun_out/prof_${T}_$cfg
  python3 scripts/summarize_pmc.py gpurun_out/${T}_$cfg $cfg $P/sq1 $P/sq2 $P/sq3 $P/tcc $P/tcp $P/fetch $P/write $P/grbm > gpurun_out/${T}_${cfg}_summary.txt 2>&1
  mv gpurun_out/${T}_${cfg}_hbm_traffic.json gpurun_out/${T}_hbm_traffic_$cfg.json
  mv gpurun_out/${T}_${cfg}_pmc_counters.csv gpurun_out/${T}_pmc_counters_$cfg.csv
  # (bench.py runs a child bench for `fast_class`: the parent's statistics file is the larger one)
  cp "$(ls -S $P/stats/*/*kernel_stats.csv $P/stats/*kernel_stats.csv 2>/dev/null | head -1)" gpurun_out/${T}_bench_kernel_stats_$cfg.csv
  # (bench.py runs a child bench for `fast_class`: the parent's statistics file is the larger one)
  cp "$(ls -S $P/stats_if1/*/*kernel_stats.csv $P/stats_if1/*kernel_stats.csv 2>/dev/null | head -1)" gpurun_out/${T}_bench_kernel_stats_inflight1_$cfg.csv
  tail -12 gpurun_out/${T}_${cfg}_summary.txt | cut -c1-400
  rm -rf "$P"        # the raw traces (20 MB per config): gpurun copies back at most 64 MiB
done
ls -la gpurun_out/${T}_*
