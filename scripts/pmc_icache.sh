#!/bin/bash
# Dev aid: instruction-cache counters of the march launch (scripts/time_march.py) under the given pre-built libraries ("base" = in-tree).
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
for lib in base "$@"; do
  if [ "$lib" = base ]; then unset IFF_LIB_PATH; else export IFF_LIB_PATH="$PWD/$lib"; fi
  tag=$(basename "$lib" .so); rm -rf gpurun_out/pmc_ic_$tag
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVES --output-format csv -d gpurun_out/pmc_ic_$tag -o p -- python3 scripts/time_march.py ${CFG:-lego16k} > /dev/null 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQC_TC_INST_REQ SQC_ICACHE_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_ic2_$tag -o p -- python3 scripts/time_march.py ${CFG:-lego16k} > /dev/null 2>&1
  python3 - $tag <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for d in ("pmc_ic_", "pmc_ic2_"):
    for f in glob.glob(f"gpurun_out/{d}{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "fan_march" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print(sys.argv[1], {k: round(v[0] / max(v[1], 1)) for k, v in acc.items()})
PY
done
