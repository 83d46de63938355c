import collections, csv, sys, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("mw::(anonymous namespace)::", "").replace("void ", ""))[:50]
    if "gemm" not in n and "flash" not in n: continue
    key = (n, r.get("Grid_Size_X","?"), r.get("Grid_Size_Y","?"), r.get("Grid_Size_Z","?"))
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k, {c: round(sum(v)/len(v)) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
