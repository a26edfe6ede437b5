#!/bin/bash
# Dev aid: counters of the ViT kernels on scripts/time_vit32.py (32 images per forward, one stream): kernel durations + --pmc passes.
#     FORMS=2 PRECS=fp32 bash scripts/pmc_vit32.sh <tag>
set -u
TAG=${1:-vit32}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
export ONLY32=1
run() { local name=$1; shift
  timeout -k 10 300 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o m -- python3 scripts/time_vit32.py > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?"; }
run kt --kernel-trace --stats
run sq1 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
run sq2 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS
run tcc --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run fetch --kernel-trace --pmc FETCH_SIZE
run write --kernel-trace --pmc WRITE_SIZE
run grbm --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT
python3 - "$OUT" > gpurun_out/pmc_$TAG.txt <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
dur = {}
for fn in glob.glob(f"{out}/kt/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(fn)))[:12]:
        name = r["Name"].replace("(anonymous namespace)::", "")
        dur[name[:60]] = float(r["AverageNs"]) / 1e3
        print(name[:70].ljust(70), r["Calls"].rjust(5), "%8.1f us" % (float(r["AverageNs"]) / 1e3), r["Percentage"])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for name in ("sq1", "sq2", "tcc", "fetch", "write", "grbm"):
    for fn in glob.glob(f"{out}/{name}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(fn)):
            acc[row["Kernel_Name"].replace("(anonymous namespace)::", "")[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    if "k_vit" not in k:
        continue
    c = {n: sum(v) / len(v) for n, v in d.items()}
    us = dur.get(k)
    print("\n" + k, "avg us", us)
    print("   ", {n: round(v) for n, v in c.items()})
    if us and "SQ_WAVE_CYCLES" in c:
        cyc = us * 1e-6 * 2.4e9
        print("    derived: mfma_busy %.3f  lds_busy %.3f (conflict share %.3f)  wait_any/wave_cycles %.2f  wait_inst/wave_cycles %.2f  active/wave_cycles %.2f  waves %d"
              % (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / cyc, c.get("SQ_LDS_IDX_ACTIVE", 0) / 256 / cyc,
                 c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1), c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"],
                 c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"], c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"], c.get("SQ_WAVES", 0)))
        if "TCC_REQ_sum" in c:
            print("    L2: req %.3g hit rate %.3f  EA rdreq %.3g   FETCH_SIZE %.4g  WRITE_SIZE %.4g  (raw units)  clock %.2f GHz"
                  % (c["TCC_REQ_sum"], c.get("TCC_HIT_sum", 0) / max(c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0), 1), c.get("TCC_EA0_RDREQ_sum", 0),
                     c.get("FETCH_SIZE", 0), c.get("WRITE_SIZE", 0), c.get("GRBM_GUI_ACTIVE", 0) / 8 / (us * 1e-6) / 1e9))
PY
cat gpurun_out/pmc_$TAG.txt | cut -c1-420
rm -rf "$OUT"/*/
