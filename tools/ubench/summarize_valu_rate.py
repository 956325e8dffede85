#!/usr/bin/env python3
"""Makes the committed table of tools/ubench/valu_rate.hip's output (one JSON object per line) and of the node
step's instruction histogram (count_loop.py on the microbenchmark's assembly).

    summarize_valu_rate.py RUN.jsonl LOOP.json OUT.json
"""
import json
import sys


def main():
    rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
    loop = json.load(open(sys.argv[2]))
    names = []
    for r in rows:
        if r["row"] not in names:
            names.append(r["row"])
    table = {}
    for n in names:
        per_w = {}
        for r in rows:
            if r["row"] == n:
                per_w[str(r["waves_per_simd"])] = round(r.get("cycles_per_vector_instruction_and_simd", r["cycles_per_turn_and_simd"]), 3)
        note = next(r["note"] for r in rows if r["row"] == n)
        table[n] = {"cycles_per_instruction_and_simd_by_waves_per_simd": per_w, "note": note}
    node = [r for r in rows if r["row"].startswith("node step")]
    out = {
        "what": "gfx950 (MI355X) vector issue rates measured in shader cycles (s_memtime) inside the kernel, waves per SIMD "
                "checked from HW_ID; tools/ubench/valu_rate.hip",
        "clock_GHz_median": round(sorted(r["clock_GHz"] for r in rows)[len(rows) // 2], 3),
        "placement_checked": all(r["waves_per_simd_placed"] == [r["waves_per_simd"]] * 2 and r["simds_seen"] == 1024 for r in rows),
        "rows": table,
        "node_step": {
            "function": "rtd::qnode_children_sorted (rt_traverse.hpp) on register-resident node words, 64 lanes active, no memory access",
            "loop_instructions": {k: loop[k] for k in ("vector", "scalar", "nop_or_wait", "lds", "memory")},
            "vector_histogram": loop["vector_histogram"],
            "cycles_per_step_and_simd_by_waves_per_simd": {str(r["waves_per_simd"]): round(r["cycles_per_turn_and_simd"], 1) for r in node},
            "cycles_per_vector_instruction_and_simd_by_waves_per_simd": {
                str(r["waves_per_simd"]): round(r["cycles_per_vector_instruction_and_simd"], 3) for r in node},
        },
    }
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out["node_step"]["cycles_per_vector_instruction_and_simd_by_waves_per_simd"]))


if __name__ == "__main__":
    main()
