#!/usr/bin/env python3
"""FETCH_SIZE calibration: bytes the gather microbenchmark is known to read / bytes the counter reports.

    python tools/fetch_calibration.py <dir with the rocprofv3 --pmc FETCH_SIZE csv> <stdout of gather_fetch> <out.json>

FETCH_SIZE is in KiB.  The factor for `gather128` (one 128-byte line per lane at random, the node fetch of the
closest-hit kernel) is what bench.py multiplies the kernel's FETCH_SIZE with."""
import collections
import csv
import glob
import json
import os
import sys

d, stdout_path, out_path = sys.argv[1:4]
known = {}
for line in open(stdout_path):
    line = line.strip()
    if line.startswith("{"):
        r = json.loads(line)
        known[r["kernel"]] = r  # the last repetition
sums, cnt = collections.defaultdict(float), collections.Counter()
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            k = r["Kernel_Name"].split("(")[0]
            sums[k] += float(r["Counter_Value"])
            cnt[k] += 1
res = {"note": "factor = known bytes / (FETCH_SIZE x 1024); table far larger than the Infinity Cache, random records"}
for k, r in known.items():
    if cnt.get(k):
        reported = sums[k] / cnt[k] * 1024.0
        res[k] = {
            "fetch_size_bytes": reported,
            "bytes_used": r["bytes_used"],
            "bytes_as_whole_128B_lines": r["bytes_lines"],
            "factor_vs_lines": r["bytes_lines"] / reported,
            "factor_vs_used": r["bytes_used"] / reported,
            "ms": r["ms"],
            "GBps_lines": r["bytes_lines"] / r["ms"] / 1e6,
        }
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res, indent=1))
