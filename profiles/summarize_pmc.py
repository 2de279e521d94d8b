#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/..., scratch) into the small, committed files under profiles/.

    python profiles/summarize_pmc.py r03 gpurun_out/prof_r03_train gpurun_out/pmc_r03_train_FETCH_SIZE gpurun_out/pmc_r03_train_WRITE_SIZE ...
    (normally through  bash profiles/collect.sh --summarize r03)

  <tag>_kernel_stats.csv   verbatim `rocprofv3 --kernel-trace --stats` per-kernel table
  <tag>_pmc_summary.json   mean counter value per dispatch and kernel; FETCH_SIZE / WRITE_SIZE (KB) converted to
                           bytes, FETCH doubled as MI355X_MICROARCH.md §HBM prescribes for gfx950.
Counters are collected in their own runs (one --pmc pass per TCC counter) with no tracing alongside.
"""
import collections
import csv
import glob
import json
import os
import sys


def newest(pattern):
    """gpurun merges every call's files into the same scratch directory: take the most recent run only."""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:]


def load(d):
    rows = []
    for f in newest(os.path.join(d, "*", "*counter_collection.csv")):
        rows += list(csv.DictReader(open(f)))
    return rows


def main():
    tag, stats_dir, pmc_dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    here = os.path.dirname(os.path.abspath(__file__))
    src = newest(os.path.join(stats_dir, "*", "*kernel_stats.csv"))[0]
    open(os.path.join(here, f"{tag}_kernel_stats.csv"), "w").write(open(src).read())
    summ = collections.defaultdict(dict)
    for d in pmc_dirs:
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in load(d):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            for c, v in cs.items():
                summ[k][c + "_mean_per_dispatch"] = sum(v) / len(v)
                summ[k][c + "_dispatches"] = len(v)
    out = {}
    for k, d in summ.items():
        if "anonymous namespace" not in k:
            continue
        if "FETCH_SIZE_mean_per_dispatch" in d:
            d["hbm_read_bytes_corrected"] = d["FETCH_SIZE_mean_per_dispatch"] * 1024 * 2
        if "WRITE_SIZE_mean_per_dispatch" in d:
            d["hbm_write_bytes"] = d["WRITE_SIZE_mean_per_dispatch"] * 1024
        h, m = d.get("TCC_HIT_sum_mean_per_dispatch"), d.get("TCC_MISS_sum_mean_per_dispatch")
        if h is not None and m is not None and h + m > 0:
            d["l2_hit_rate"] = h / (h + m)
        out[k] = d
    json.dump(out, open(os.path.join(here, f"{tag}_pmc_summary.json"), "w"), indent=1, sort_keys=True)
    print("wrote", tag, len(out), "kernels")


if __name__ == "__main__":
    main()
