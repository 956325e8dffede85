#!/usr/bin/env python3
"""Condenses a tools/profile_extend.sh directory into one JSON: per-launch figures of the closest-hit kernel
(uninstrumented launches only) and of the shade kernel, stamped with the hash of the kernel sources and the
bench arguments they were taken with, so that bench.py can tell whether a committed profile still describes
the code it runs.

FETCH_SIZE is in KiB of fabric read requests; `hbm_bytes_per_launch_raw` = FETCH_SIZE x 1024.  What factor turns
it into bytes for this kernel's access pattern is measured separately (tools/ubench/gather_fetch.hip,
tools/fetch_calibration.py); bench.py applies it."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from srchash import kernel_source_hash  # noqa: E402

d = sys.argv[1]
out_path = sys.argv[2]
KERNELS = {"extend": "extend_kernel<false", "shade": "shade_kernel"}


def rows(pattern):
    for f in glob.glob(os.path.join(d, pattern), recursive=True):
        yield from csv.DictReader(open(f))


trace = list(rows("trace/**/*kernel_trace.csv"))
pmc = list(rows("pmc_*/**/*counter_collection.csv"))


def summarise(name):
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in trace if name in r["Kernel_Name"]]
    if not dur:
        return None
    sums, cnt = collections.defaultdict(float), collections.Counter()
    for r in pmc:
        if name in r["Kernel_Name"]:
            sums[r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[r["Counter_Name"]] += 1
    n = len(dur)
    res = {"kernel": name, "launches": n, "mean_launch_ms": sum(dur) / n / 1e6, "total_ms": sum(dur) / 1e6}
    for r in trace:
        if name in r["Kernel_Name"]:
            res["vgpr"], res["sgpr"], res["lds_bytes"] = r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size")
            try:
                # 256-thread blocks (one wave per SIMD each): as many per CU as the 160 KiB of LDS hold, 8 at most
                res["waves_per_simd"] = min(8, 163840 // max(1, int(res["lds_bytes"]))) if int(res["lds_bytes"]) else 8
            except (TypeError, ValueError):
                pass
            break
    if cnt.get("FETCH_SIZE"):
        res["hbm_bytes_per_launch_raw"] = sums["FETCH_SIZE"] * 1024.0 / cnt["FETCH_SIZE"]
    if cnt.get("SQ_ACTIVE_INST_VALU") and cnt.get("SQ_THREAD_CYCLES_VALU"):
        res["valu_lane_utilisation"] = sums["SQ_THREAD_CYCLES_VALU"] / (sums["SQ_ACTIVE_INST_VALU"] * 64.0)
    if cnt.get("GRBM_GUI_ACTIVE") and cnt.get("SQ_ACTIVE_INST_VALU"):
        # GRBM_GUI_ACTIVE sums the 8 XCDs; SQ_ACTIVE_INST_* count quad-cycles over 1024 SIMDs
        res["valu_busy_fraction"] = sums["SQ_ACTIVE_INST_VALU"] / cnt["SQ_ACTIVE_INST_VALU"] * 4.0 / (
            sums["GRBM_GUI_ACTIVE"] / cnt["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        res["clock_GHz_under_profile"] = sums["GRBM_GUI_ACTIVE"] / cnt["GRBM_GUI_ACTIVE"] / 8.0 / (sum(dur) / n)
    if cnt.get("TCC_HIT_sum"):
        res["l2_hit_rate"] = sums["TCC_HIT_sum"] / (sums["TCC_HIT_sum"] + sums["TCC_MISS_sum"])
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
              "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT",
              "TCC_REQ_sum", "TCC_READ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_READ_sum",
              "TCP_PENDING_STALL_CYCLES_sum", "TA_TA_BUSY_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TD_TD_BUSY_sum",
              "TD_TC_STALL_sum"):
        if cnt.get(k):
            res[k + "_per_launch"] = sums[k] / cnt[k]
    return res


out = {"source_hash": kernel_source_hash()}
if os.environ.get("RT_GIT_HEAD"):
    out["git_head"] = os.environ["RT_GIT_HEAD"]
args_file = os.path.join(d, "bench_args.txt")
if os.path.exists(args_file):
    out["bench_args"] = open(args_file).read().strip()
for key, name in KERNELS.items():
    s = summarise(name)
    if s:
        out[key] = s
bl = os.path.join(d, "bench_line_under_rocprof.json")
if os.path.exists(bl):
    try:
        out["bench_line_under_rocprof"] = json.loads(open(bl).read().strip().splitlines()[-1])
    except Exception:
        pass
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "bench_line_under_rocprof"}, indent=1))
