#!/usr/bin/env python3
"""Makes the committed table of tools/ubench/quad_fetch.hip's output (one JSON object per measurement).

    summarize_gather_rate.py RUN.txt OUT.json

What bench.py reads from the table: the rate of per-lane 16-byte loads that hit in the vector L1 (accesses per clock
and CU), and the rates at which a CU's L1 misses are filled from the L2 and from beyond it (lines per second, whole chip).
"""
import json
import sys

CLOCK_GHZ = 2.4  # only used to express a rate as clocks per node; the rates themselves are wall-clock


def main():
    rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
    cus = rows[0]["cus"]

    def rate(mode, mb, pc, blocks=7):
        for r in rows:
            if r["mode"] == mode and r["table_MB"] == mb and r["active_percent"] == pc and r["blocks_per_cu"] == blocks:
                return r["G_nodes_per_s"]
        return None

    def clk(g):
        return round(cus * CLOCK_GHZ / g, 3) if g else None

    sizes = sorted({r["table_MB"] for r in rows})
    small = sizes[0]
    table = []
    for mb in sizes:
        for pc in (100, 60):
            entry = {"table_MB": mb, "active_percent": pc}
            for mode, name in ((0, "4x16B"), (2, "3x16B"), (3, "2x16B"), (4, "8x8B"), (5, "4x12B"), (6, "16x4B"), (1, "4x16B_one_node_per_quad")):
                g = rate(mode, mb, pc)
                entry[name] = {"G_nodes_per_s": g, "clk_per_node_and_cu": clk(g)}
            table.append(entry)
    l2_sizes = [mb for mb in sizes if 1 <= mb <= 4]
    out = {
        "what": "gfx950 (MI355X): every lane of a wave gathers one 64-byte-aligned node of its own from a table of the given "
                "size, 7 blocks of 256 threads per CU (tools/ubench/quad_fetch.hip); rates are whole-chip, wall clock",
        "cus": cus,
        "clock_GHz_assumed_for_clk_columns": CLOCK_GHZ,
        "l1_hit": {
            "table_MB": small,
            "accesses_per_clk_and_cu_all_lanes": round(4.0 / clk(rate(0, small, 100)), 3),
            "accesses_per_clk_and_cu_60_percent_lanes": round(4.0 / clk(rate(0, small, 60)), 3),
            "bytes_per_clk_and_cu_all_lanes": round(64.0 / clk(rate(0, small, 100)), 2),
            "note": "four 16-byte loads per lane and node, every lane another line, table resident in the L1: the cost "
                    "follows the bytes (3 x 16, 4 x 12 and 2 x 16 bytes cost 3/4, 3/4 and 1/2; 8 x 8 and 16 x 4 bytes cost the same)",
        },
        "line_fill_G_per_s": {
            "from_l2": round(max(rate(3, mb, 100) for mb in l2_sizes), 1),
            "from_l2_note": "2 x 16 bytes per lane (the L1's own rate does not bound it), tables of 1 to 4 MB",
            "beyond_l2_16MB_table": rate(0, 16, 100),
            "beyond_l2_2GB_table": rate(0, 2048, 100),
            "beyond_l2_note": "one 128-byte line per node whatever the bytes loaded from it; 53 G lines/s = 6.8 TB/s",
        },
        "one_node_per_quad": {
            "note": "the four lanes of a quad fetch one node per load (one L1 access per node instead of four), then a 4 x 4 "
                    "transposition of 32 v_cndmask_b32_dpp: faster only with all lanes active and the table in the L1",
            "clk_per_node_and_cu": {"%g MB, %d %% lanes" % (mb, pc): {"per_lane": clk(rate(0, mb, pc)), "per_quad": clk(rate(1, mb, pc))}
                                    for mb in (small, 2, 16, 2048) for pc in (100, 60)},
        },
        "rows": table,
    }
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("l1_hit", "line_fill_G_per_s")}, indent=1))
    print(json.dumps(out["one_node_per_quad"]["clk_per_node_and_cu"], indent=1))


if __name__ == "__main__":
    main()
