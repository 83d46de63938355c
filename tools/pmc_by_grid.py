#!/usr/bin/env python3
"""Development tool: per-(kernel, grid) averages of every counter in a rocprofv3 --pmc counter_collection.csv, for the GEMM and
fused-attention kernels of the encoder (usage: pmc_by_grid.py <counter_collection.csv>; see profiles/r01_pmc_encoder_mfma.txt)."""
import collections, csv, sys, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("mw::(anonymous namespace)::", "").replace("void ", ""))[:50]
    if "gemm" not in n and "flash" not in n: continue
    key = (n, "threads=" + r.get("Grid_Size", "?"), "wg=" + r.get("Workgroup_Size", "?"))
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    avg = {c: sum(v) / len(v) for c, v in d.items()}
    util = ""
    if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and avg.get("GRBM_GUI_ACTIVE", 0) > 0:
        # busy cycles summed over the 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
        util = " mfma_util=%5.1f%%" % (100.0 * avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * avg["GRBM_GUI_ACTIVE"] / 8.0))
    print(k, "n=%d" % len(next(iter(d.values()))), util, {c: round(v) for c, v in avg.items()})
