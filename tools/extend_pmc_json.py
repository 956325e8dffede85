#!/usr/bin/env python3
"""Condenses a tools/profile_extend.sh directory into one JSON: per-launch figures of the closest-hit
kernel (uninstrumented launches only) for bench.py's roofline.traffic.
FETCH_SIZE is in KiB of fabric read requests tallied at 64 B; on gfx950 a wide (16 B per lane) read
stream reports half of its bytes, so `hbm_bytes_corrected` doubles it.  That factor is calibrated for
coalesced streams, not for this kernel's one-line-per-lane gathers; both figures are given."""
import collections
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
out_path = sys.argv[2]
name = "extend_kernel<false>"


def rows(pattern):
    for f in glob.glob(os.path.join(d, pattern), recursive=True):
        yield from csv.DictReader(open(f))


dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows("trace/**/*kernel_trace.csv") if name in r["Kernel_Name"]]
sums, cnt = collections.defaultdict(float), collections.Counter()
for r in rows("pmc_*/**/*counter_collection.csv"):
    if name in r["Kernel_Name"]:
        sums[r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[r["Counter_Name"]] += 1
n = len(dur)
res = {
    "kernel": "extend_kernel (closest hit), uninstrumented launches",
    "launches": n,
    "mean_launch_ms": sum(dur) / n / 1e6,
    "total_ms": sum(dur) / 1e6,
}
if cnt.get("FETCH_SIZE"):
    raw = sums["FETCH_SIZE"] * 1024.0 / cnt["FETCH_SIZE"]
    res["hbm_bytes_per_launch_raw"] = raw
    res["hbm_bytes_per_launch_corrected"] = 2.0 * raw
if cnt.get("SQ_ACTIVE_INST_VALU") and cnt.get("SQ_THREAD_CYCLES_VALU"):
    res["valu_lane_utilisation"] = sums["SQ_THREAD_CYCLES_VALU"] / (sums["SQ_ACTIVE_INST_VALU"] * 64.0)
if cnt.get("GRBM_GUI_ACTIVE") and cnt.get("SQ_ACTIVE_INST_VALU"):
    # GRBM_GUI_ACTIVE sums the 8 XCDs; SQ_ACTIVE_INST_* count quad-cycles over 1024 SIMDs
    res["valu_busy_fraction"] = sums["SQ_ACTIVE_INST_VALU"] / cnt["SQ_ACTIVE_INST_VALU"] * 4.0 / (
        sums["GRBM_GUI_ACTIVE"] / cnt["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
if cnt.get("TCC_HIT_sum"):
    res["l2_hit_rate"] = sums["TCC_HIT_sum"] / (sums["TCC_HIT_sum"] + sums["TCC_MISS_sum"])
for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
          "SQ_ACTIVE_INST_ANY"):
    if cnt.get(k):
        res[k + "_per_launch"] = sums[k] / cnt[k]
bl = os.path.join(d, "bench_line_under_rocprof.json")
if os.path.exists(bl):
    try:
        res["bench_line_under_rocprof"] = json.loads(open(bl).read().strip().splitlines()[-1])
    except Exception:
        pass
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "bench_line_under_rocprof"}, indent=1))
