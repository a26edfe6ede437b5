#!/bin/bash
# PMC passes over scripts/time_march.py (the march alone).  usage: bash scripts/pmc_march.sh <tag> [config]
set -u
TAG=${1:-fan}; CFG=${2:-lego16k}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG; mkdir -p "$OUT"
run() { local name=$1; shift
  timeout -k 10 300 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o m -- python3 scripts/time_march.py $CFG > "$OUT/$name.log" 2>&1
  local rc=$?; echo "pass $name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
run stats --kernel-trace --stats
run sq1 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
run sq2 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR
run sq3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for name in ("sq1", "sq2", "sq3"):
    files = glob.glob(f"{out}/{name}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in files:
        for row in csv.DictReader(open(fn)):
            acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in acc.items():
        if "k4" in k or "ref_shade" in k:
            print(name, k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "n", len(next(iter(d.values()))))
for fn in glob.glob(f"{out}/stats/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        if "k4" in row["Name"] or "ref_shade" in row["Name"]:
            print("stats", row["Name"][:60], row["Calls"], row["AverageNs"])
PY
