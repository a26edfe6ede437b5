#!/bin/bash
# Dev aid: a short bench + the true (in-flight 1) durations of the kernels named by the grep pattern in $1.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
pat=${1:-"colsum|topk|merge|fan_march|trunk_h<1"}
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-instrument 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('poses/s', j['value'], 'ms/step', j['ms_per_step'], 'warm', j.get('warm_poses_per_s'))" || exit 1
rm -rf gpurun_out/prof_ks
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ks -o ks -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-instrument --in-flight 1 > /dev/null 2>&1
f=$(find gpurun_out/prof_ks -name "*kernel_stats.csv" | head -1)
python - "$f" "$pat" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]): print(r["Name"][:60], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
