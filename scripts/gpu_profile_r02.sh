#!/bin/bash
# Round-2 evidence run on the GPU box: rocprofv3 passes (scripts/profile_pmc.sh), their summary, the full-size parity record.
set -e
bash scripts/profile_pmc.sh r02 > gpurun_out/profile_r02.log 2>&1 || { tail -20 gpurun_out/profile_r02.log; exit 1; }
P=gpurun_out/prof_r02
python3 scripts/summarize_pmc.py gpurun_out/r02 lego16k $P/sq1 $P/sq2 $P/sq3 $P/tcc $P/tcp $P/fetch $P/write $P/grbm
cp "$(ls $P/stats/*/*kernel_stats.csv $P/stats/*kernel_stats.csv 2>/dev/null | head -1)" gpurun_out/r02_bench_kernel_stats.csv
cp "$(ls $P/stats_if1/*/*kernel_stats.csv $P/stats_if1/*kernel_stats.csv 2>/dev/null | head -1)" gpurun_out/r02_bench_kernel_stats_inflight1.csv
tail -3 gpurun_out/profile_r02.log
ls -la gpurun_out/r02_*
