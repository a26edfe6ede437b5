#!/bin/bash
# Dev aid: per-kernel durations of a 32-image ViT forward (scripts/time_vit32.py, one stream) for each GEMM form of iff_vit_desc.gemm_form.
#     bash scripts/gpu_vit_forms.sh "1 2 4" [fp32|bf16]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
PRECS=${2:-fp32}
for f in ${1:-1 2}; do
  OUT=gpurun_out/vitform_$f; rm -rf "$OUT"
  FORMS=$f PRECS=$PRECS ONLY32=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o m -- python3 scripts/time_vit32.py > gpurun_out/vitform_$f.log 2>&1 || { tail -5 gpurun_out/vitform_$f.log; exit 1; }
  grep vit_ms gpurun_out/vitform_$f.log
  python3 - "$OUT" <<'PY'
import csv, glob, sys
for fn in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(fn)))[:9]:
        print("   ", r["Name"].replace("(anonymous namespace)::", "")[:78].ljust(78), r["Calls"].rjust(5), "%8.1f us" % (float(r["AverageNs"]) / 1e3), r["Percentage"])
PY
  rm -rf "$OUT"
done
