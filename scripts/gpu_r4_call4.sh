#!/bin/bash
# round 4, call 4: GPU suite (merge kernels, fp64 referee, RCCL rehearsal)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
echo "== pytest $(date +%T)"
timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=6 > gpurun_out/c4_pytest.log 2>&1; rc=$?; tail -n 30 gpurun_out/c4_pytest.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
cat gpurun_out/rccl_rehearsal.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/fullsize_parity.json"))
print({k:v for k,v in d.get("lego16k",{}).items() if "near_tie" in k or "top100_lists" in k})
PY
echo "== done $(date +%T)"
