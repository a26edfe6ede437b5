#!/bin/bash
# Dev aid: image -> pose throughput (QS images per graph, 4 graphs in flight) and the per-kernel durations of one graph at a time.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
QS=${QS:-32}
QS=$QS timeout -k 10 300 python scripts/time_image_to_pose.py 2>/dev/null | grep images_per || exit 1
rm -rf gpurun_out/prof_vit
QS=$QS INFLIGHT=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_vit -o ks -- python3 scripts/time_image_to_pose.py > /dev/null 2>&1
f=$(find gpurun_out/prof_vit -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY' > gpurun_out/ks_vit${TAG}.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r["Name"][:80], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), r["Percentage"])
PY
head -${LINES_SHOWN:-12} gpurun_out/ks_vit${TAG}.txt
