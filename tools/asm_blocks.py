#!/usr/bin/env python3
"""Basic blocks of one kernel in a gfx950 assembly file: instruction counts per block and where each block branches.

    asm_blocks.py FILE.s KERNEL_SUBSTRING [--dump LABEL ...]
"""
import re
import sys


def main():
    path, sub = sys.argv[1], sys.argv[2]
    dump = sys.argv[4:] if len(sys.argv) > 3 and sys.argv[3] == "--dump" else []
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and sub in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur = [], None
    for l in lines[start:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m or cur is None:
            cur = {"name": m.group(1) if m else "entry", "v": 0, "s": 0, "vm": 0, "ds": 0, "scr": 0, "wait": 0, "br": [], "text": []}
            blocks.append(cur)
            if m:
                continue
        t = l.strip().split()
        if not t or t[0].startswith(";") or t[0].startswith("."):
            continue
        cur["text"].append(l.rstrip()[:110])
        op = t[0]
        if op.startswith("v_"):
            cur["v"] += 1
        elif op in ("s_waitcnt", "s_nop"):
            cur["wait"] += 1
        elif op.startswith("s_cbranch") or op == "s_branch":
            cur["br"].append(op.replace("s_cbranch_", "") + ">" + t[1])
            cur["s"] += 1
        elif op.startswith("s_"):
            cur["s"] += 1
        elif op.startswith("global_") or op.startswith("flat_"):
            cur["vm"] += 1
        elif op.startswith("scratch_") or op.startswith("buffer_"):
            cur["scr"] += 1
        elif op.startswith("ds_"):
            cur["ds"] += 1
    for b in blocks:
        print("%-12s v=%3d s=%3d vm=%2d ds=%2d scr=%d wait/nop=%d %s" % (b["name"], b["v"], b["s"], b["vm"], b["ds"], b["scr"], b["wait"], " ".join(b["br"])))
    for name in dump:
        for b in blocks:
            if b["name"] == name:
                print("==== " + name)
                print("\n".join(b["text"]))


if __name__ == "__main__":
    main()
