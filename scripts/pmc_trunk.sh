#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_trunk; mkdir -p "$OUT"
run() { local name=$1; shift
  timeout -k 10 300 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o m -- python3 scripts/coresidency.py > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?"; }
run sq1 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
run sq2 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_SALU
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for name in ("sq1", "sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in glob.glob(f"{out}/{name}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(fn)):
            acc[row["Kernel_Name"][:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in acc.items():
        if "k5_trunk" in k:
            print(name, k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "n", len(next(iter(d.values()))))
PY
