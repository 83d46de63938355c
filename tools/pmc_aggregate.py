#!/usr/bin/env python3
"""Per-kernel average / min / max of one counter of a rocprofv3 --pmc counter_collection.csv (the raw file is too
large to keep).  usage: pmc_aggregate.py <counter_collection.csv> <COUNTER_NAME> <out.csv>"""
import collections
import csv
import sys


def main():
    src, counter, out = sys.argv[1:4]
    agg = collections.defaultdict(lambda: [0, 0.0, float("inf"), 0.0])
    for r in csv.DictReader(open(src)):
        if r["Counter_Name"] != counter:
            continue
        v = float(r["Counter_Value"])
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += v
        a[2] = min(a[2], v)
        a[3] = max(a[3], v)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Dispatches", f"{counter}_KiB_avg", f"{counter}_KiB_min", f"{counter}_KiB_max"])
        for name, (n, tot, lo, hi) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([name, n, round(tot / n, 2), lo, hi])


if __name__ == "__main__":
    main()
