#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel: sum and mean per dispatch.
usage: pmc_summary.py <dir> [kernel-substring]"""
import csv, glob, os, sys, collections
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            if pat and pat not in k:
                continue
            k = k.split("(")[0][:60]
            c = row["Counter_Name"]
            acc[k][c] += float(row["Counter_Value"])
            cnt[k][c] += 1
for k in acc:
    for c in acc[k]:
        print(f"{k:60s} {c:14s} dispatches {cnt[k][c]:7d} sum {acc[k][c]:.6e} mean {acc[k][c]/cnt[k][c]:.6e}")
