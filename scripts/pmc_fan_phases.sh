#!/bin/bash
# Vector instructions of the fan kernel phase by phase: builds that stop every tile after a phase (build/lib_exit<k>.so, made with
#   bash scripts/build_one_tu.sh exit<k> fan_march_kernels.hip -DFAN_EXIT_AFTER=<k>) under one --pmc pass each; dev aid.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
for k in 1 2 3 5 7 9 11 13 full; do
  if [ $k = full ]; then unset IFF_LIB_PATH; else export IFF_LIB_PATH="$PWD/build/lib_exit$k.so"; fi      # never copied over the product library
  rm -rf gpurun_out/pmc_exit$k
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/pmc_exit$k -o p -- python3 scripts/time_march.py > /dev/null 2>&1
  python3 - $k <<'PY'
import csv, glob, sys, collections
f = glob.glob(f"gpurun_out/pmc_exit{sys.argv[1]}/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f[0])):
    if "fan_march" in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print("exit after", sys.argv[1], {k: round(v[0] / max(v[1], 1)) for k, v in acc.items()})
PY
done
