#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes into profiles/: per-kernel means of FETCH_SIZE / WRITE_SIZE (KiB per dispatch) and the
HBM-side bytes per launch with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE under-reports by 2x):

    hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024

usage: summarize_pmc.py <fetch_dir> <write_dir> <out_prefix>      e.g.  gpurun_out/pmc_fetch2 gpurun_out/pmc_write2 profiles/r01v3
"""
import csv, glob, json, re, sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:120]


def means(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v), min(v), max(v)) for k, v in acc.items()}


def main():
    fd, wd, out = sys.argv[1:4]
    fe, wr = means(fd, "FETCH_SIZE"), means(wd, "WRITE_SIZE")
    for tag, m in (("FETCH_SIZE", fe), ("WRITE_SIZE", wr)):
        with open(f"{out}_pmc_{tag}.csv", "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["Kernel_Name", "Counter", "Dispatches", "Mean_KB", "Min_KB", "Max_KB"])
            for k in sorted(m):
                n, mean, lo, hi = m[k]
                w.writerow([k, tag, n, round(mean, 1), lo, hi])
    js = {}
    for k in sorted(set(fe) | set(wr)):
        f = fe.get(k, (0, 0.0, 0, 0))[1]
        w = wr.get(k, (0, 0.0, 0, 0))[1]
        js[k] = {"fetch_KiB_raw": round(f, 1), "write_KiB": round(w, 1), "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
    json.dump(js, open(f"{out}_hbm_traffic.json", "w"), indent=1, sort_keys=True)
    for k in ("k4b_appearance<27, true, 1>", "k4b_appearance<27, true>", "k4b_appearance<27>", "k4a_density_composite<1>", "k4a_density_composite", "k_ref_shade<27, true>", "k5_trunk<true, 1>", "k5_trunk<true>", "k6_colsum", "k_surface_sample"):
        if k in js:
            print(k, js[k])


if __name__ == "__main__":
    main()
