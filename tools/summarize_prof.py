#!/usr/bin/env python3
"""Summarises a tools/profile_extend.sh output directory: per-kernel time and PMC sums for the extend kernel."""
import csv, glob, os, sys, collections
d = sys.argv[1]
def find(pattern):
    return sorted(glob.glob(os.path.join(d, pattern), recursive=True))
for f in find("trace/**/*kernel_stats.csv"):
    print("== kernel stats", f)
    for i, row in enumerate(csv.reader(open(f))):
        if i < 12: print("  ", ",".join(c[:60] for c in row))
for f in find("pmc_*/**/*counter_collection.csv"):
    sums = collections.defaultdict(float); n = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "extend_kernel" not in k: continue
        sums[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
    print("== pmc", os.path.relpath(f, d))
    for c in sorted(sums): print("   %-40s sum=%.6g  dispatches=%d" % (c, sums[c], n[c]))
