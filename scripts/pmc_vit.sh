#!/bin/bash
# Dev aid: counters of the ViT kernels (scripts/time_vit.py, 16 images per forward, fp32 class) -- two --pmc passes + the kernel durations.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_vit; rm -rf "$OUT"; mkdir -p "$OUT"
run() { local name=$1; shift
  timeout -k 10 300 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o m -- python3 scripts/time_vit.py > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?"; }
run kt --kernel-trace --stats
run sq1 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
run sq2 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for fn in glob.glob(f"{out}/kt/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(fn)))[:14]:
        print(r["Name"][:64].ljust(64), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us", r["Percentage"])
for name in ("sq1", "sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in glob.glob(f"{out}/{name}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(fn)):
            acc[row["Kernel_Name"][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in acc.items():
        if "k_vit" in k:
            print(name, k[24:70], {c: round(sum(v) / len(v)) for c, v in d.items()}, "n", len(next(iter(d.values()))))
PY
