#!/usr/bin/env python3
"""Instruction counts of a loop in a gfx950 assembly file (hipcc -save-temps).

    count_loop.py FILE.s KERNEL_SUBSTRING [--markers A B]

Finds the kernel whose label contains KERNEL_SUBSTRING, then the loop that holds the marker `s_nop A`
(default 3; the microbenchmark brackets the measured call with `s_nop 3` / `s_nop 4`): the innermost span from a
label to a backward branch to that label that contains the marker.  Prints JSON: vector, scalar (markers, s_nop
and s_waitcnt excluded), memory and LDS instruction counts of the loop and a histogram of its vector opcodes.
"""
import collections
import json
import re
import sys


def main():
    path, sub = sys.argv[1], sys.argv[2]
    marker = "s_nop 3"
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and sub in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    at = next(i for i, l in enumerate(body) if l.strip() == marker)
    best = None
    for i, l in enumerate(body):
        t = l.strip().split()
        if len(t) >= 2 and (t[0].startswith("s_cbranch") or t[0] == "s_branch") and t[1] in labels:
            head = labels[t[1]]
            if head <= at <= i and (best is None or i - head < best[1] - best[0]):
                best = (head, i)
    if best is None:
        raise SystemExit("no loop around the marker")
    hist = collections.Counter()
    n = collections.Counter()
    for l in body[best[0]:best[1] + 1]:
        t = l.strip().split()
        if not t or t[0].startswith(";") or t[0].startswith(".") or t[0].endswith(":"):
            continue
        op = t[0]
        if op.startswith("v_"):
            n["vector"] += 1
            hist[re.sub(r"_e(32|64)$", "", op)] += 1
        elif op in ("s_nop", "s_waitcnt"):
            n["nop_or_wait"] += 1
        elif op.startswith("s_"):
            n["scalar"] += 1
        elif op.startswith("ds_"):
            n["lds"] += 1
        else:
            n["memory"] += 1
    print(json.dumps({"kernel": sub, "loop_lines": [best[0], best[1]], "vector": n["vector"], "scalar": n["scalar"],
                      "nop_or_wait": n["nop_or_wait"], "lds": n["lds"], "memory": n["memory"],
                      "vector_histogram": dict(hist.most_common())}))


if __name__ == "__main__":
    main()
