#!/bin/bash
# rocprofv3 passes behind profiles/r02_*: kernel trace + stats, then one --pmc pass per counter group (the pool refuses
# --pmc together with the hip/hsa trace domains; --kernel-trace is allowed).  Run on the GPU box from the repo root:
#     bash scripts/profile_pmc.sh <tag> [bench args...]
# The program after `--` is python3 itself (no env / bash -c hop: the profiler's preloaded library has initialised the GPU).
set -u
TAG=${1:-r02}; shift || true
ARGS=("$@")
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1
run() {   # name, rocprof args...
    local name=$1; shift
    echo "== pass $name $(date +%T)"
    timeout -k 10 400 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o b -- python3 bench.py --steps 6 --warmup 2 --settle 0 --no-cpu-baseline ${BENCH_EXTRA:-} "${ARGS[@]}" > "$OUT/$name.log" 2>&1
    local rc=$?
    echo "   rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $name timed out: stopping"; exit $rc; fi
}
BENCH_EXTRA="--steps 60 --warmup 10" run stats --kernel-trace --stats
BENCH_EXTRA="--steps 60 --warmup 10 --in-flight 1 --no-extras" run stats_if1 --kernel-trace --stats
export BENCH_EXTRA="--in-flight 1 --no-instrument"
run sq1 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
run sq2 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_BUSY_CYCLES
run sq3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_LDS
run tcc --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run tcp --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
run fetch --kernel-trace --pmc FETCH_SIZE
run write --kernel-trace --pmc WRITE_SIZE
run grbm --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT
ls "$OUT"
