#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one directory per pass, `--output-format csv`) into profiles/.

    summarize_pmc.py <out_prefix> <config> <pass_dir> [<pass_dir> ...]

Writes
  <out_prefix>_pmc_counters.csv   per kernel: dispatches and the mean of every collected counter (one column per counter)
  <out_prefix>_hbm_traffic.json   {"source_sha16", "config", "kernels": {name: {fetch_KiB_raw, write_KiB,
                                  hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, ...derived ratios}}}
The x2 on FETCH_SIZE is the gfx950 correction of MI355X_MICROARCH.md (section HBM): FETCH_SIZE reports half the bytes of a
wide coalesced read; for 64-B gathers it is uncalibrated, so read the figure as an upper bound of 2x the raw count.
Derived per kernel where the counters exist (SQ_* cycle counters are in quad-cycles except SQ_VALU_MFMA_BUSY_CYCLES and
SQ_BUSY_CYCLES; ratios of like units only):
  valu_active_share   = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES        share of wave lifetime with a VALU instruction executing
  wait_share          = SQ_WAIT_ANY / SQ_WAVE_CYCLES                 ... parked at s_waitcnt / barrier
  issue_stall_share   = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES            ... stalled at issue
  mfma_busy_share     = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 4 SIMDs per CU ...)  reported raw, see profiles/README.md
  l2_hit_rate         = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
  l1_hit_rate         = 1 - TCP_TCC_READ_REQ_sum / TCP_TOTAL_CACHE_ACCESSES_sum
  vmem_rd_wave_insts, valu_wave_insts, mfma_wave_insts, lds_wave_insts = SQ_INSTS_* per launch (bench.py prices the gather
                        kernels with them: 16 cycles of a CU's texture path per dwordx4 wave-load, 4 cycles of a SIMD per VALU)
  lds_active_cycles, lds_bank_conflict_cycles = SQ_LDS_IDX_ACTIVE, SQ_LDS_BANK_CONFLICT per launch, summed over the CUs
"""
import csv, glob, hashlib, json, os, re, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    return name.split("(")[0][:120]


def source_fingerprint():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "iffnerf_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(d, name), "rb") as fh:
                h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def main():
    out, config, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    acc = defaultdict(lambda: defaultdict(list))          # kernel -> counter -> values
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for k in acc for c in acc[k]})
    mean = {k: {c: sum(v) / len(v) for c, v in acc[k].items()} for k in acc}
    with open(f"{out}_pmc_counters.csv", "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Kernel_Name", "Dispatches"] + counters)
        for k in sorted(acc):
            n = max(len(v) for v in acc[k].values())
            w.writerow([k, n] + [("%.6g" % mean[k][c]) if c in mean[k] else "" for c in counters])
    js = {"source_sha16": source_fingerprint(), "config": config, "kernels": {}}
    for k in sorted(acc):
        m = mean[k]
        e = {}
        if "FETCH_SIZE" in m or "WRITE_SIZE" in m:
            f, wv = m.get("FETCH_SIZE", 0.0), m.get("WRITE_SIZE", 0.0)
            e.update(fetch_KiB_raw=round(f, 1), write_KiB=round(wv, 1), hbm_bytes_per_launch=int((2 * f + wv) * 1024))
        wc = m.get("SQ_WAVE_CYCLES")
        if wc:
            for name, c in (("valu_active_share", "SQ_ACTIVE_INST_VALU"), ("wait_share", "SQ_WAIT_ANY"),
                            ("issue_stall_share", "SQ_WAIT_INST_ANY"), ("any_inst_active_share", "SQ_ACTIVE_INST_ANY"),
                            ("lds_active_share", "SQ_ACTIVE_INST_LDS"), ("vmem_active_share", "SQ_ACTIVE_INST_VMEM")):
                if c in m:
                    e[name] = round(m[c] / wc, 4)
        for name, c in (("vmem_rd_wave_insts", "SQ_INSTS_VMEM_RD"), ("valu_wave_insts", "SQ_INSTS_VALU"),
                        ("mfma_wave_insts", "SQ_INSTS_MFMA"), ("lds_wave_insts", "SQ_INSTS_LDS"),
                        ("lds_active_cycles", "SQ_LDS_IDX_ACTIVE"), ("lds_bank_conflict_cycles", "SQ_LDS_BANK_CONFLICT"),
                        ("waves", "SQ_WAVES")):
            if c in m:
                e[name] = int(m[c])
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in m and "TCP_TCC_READ_REQ_sum" in m and m["TCP_TOTAL_CACHE_ACCESSES_sum"] > 0:
            e["l1_hit_rate"] = round(1.0 - m["TCP_TCC_READ_REQ_sum"] / m["TCP_TOTAL_CACHE_ACCESSES_sum"], 4)
        if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m and m["TCC_HIT_sum"] + m["TCC_MISS_sum"] > 0:
            e["l2_hit_rate"] = round(m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), 4)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CYCLES" in m and m["SQ_BUSY_CYCLES"] > 0:
            e["mfma_busy_cycles_over_sq_busy_cycles"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["SQ_BUSY_CYCLES"], 4)
        if e:
            js["kernels"][k] = e
    json.dump(js, open(f"{out}_hbm_traffic.json", "w"), indent=1, sort_keys=True)
    for k in js["kernels"]:
        if any(t in k for t in ("k4f", "k4g", "k4b", "k4a", "k5_trunk", "k_ref_shade", "k6_", "k_surface", "k_score")):
            print(k, js["kernels"][k])


if __name__ == "__main__":
    main()
