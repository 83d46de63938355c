#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV (development tool).

usage: trace_gaps.py <kernel_trace.csv> [min_start_fraction | steps:A:B]
(steps:A:B = from the A-th to the B-th step_begin_kernel / embed_kernel launch, i.e. decode steps A..B-1 of the run)
Prints, per (previous kernel -> next kernel) pair, the count and the mean / total gap, and the totals of busy and
idle time, for the part of the trace after `min_start_fraction` of its span (default 0.5: skips model setup).
"""
import collections
import csv
import re
import sys


def short(n):
    n = n.replace("mw::(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)[:44]


def main():
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(sys.argv[1]))]
    rows.sort()
    sel = sys.argv[2] if len(sys.argv) > 2 else "0.5"
    if sel.startswith("steps:"):
        a, b = (int(v) for v in sel.split(":")[1:3])
        emb = [i for i, r in enumerate(rows) if r[2].startswith(("step_begin_kernel", "embed_kernel"))]  # first kernel of a decode call
        rows = rows[emb[a]:emb[b]]
    else:
        t0 = rows[0][0] + float(sel) * (rows[-1][1] - rows[0][0])
        rows = [r for r in rows if r[0] >= t0]
    busy = sum(e - s for s, e, _ in rows)
    gaps = collections.defaultdict(lambda: [0, 0])
    for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
        g = gaps[(n0, n1)]
        g[0] += 1
        g[1] += max(0, s1 - e0)
    idle = sum(g[1] for g in gaps.values())
    print(f"kernels {len(rows)}  busy {busy / 1e6:.2f} ms  idle {idle / 1e6:.2f} ms  span {(rows[-1][1] - rows[0][0]) / 1e6:.2f} ms")
    for (a, b), (n, tot) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:24]:
        print(f"{a:44s} -> {b:44s} n={n:6d} mean={tot / n / 1e3:7.2f} us total={tot / 1e6:7.2f} ms")


if __name__ == "__main__":
    main()
