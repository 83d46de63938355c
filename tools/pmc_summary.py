#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one pass each, `--kernel-trace` only) of

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- python3 bench.py --steps 1 --warmup 0 \
        --no-cpu-baseline --roofline-others ""

into per-kernel HBM bytes per launch.  Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section)
prescribes for gfx950: both counters are in KiB; FETCH_SIZE tallies the 128-byte requests of wide (16 B/lane) streaming
reads at 64 bytes, so it is doubled; WRITE_SIZE is taken as is.

usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
"""
import collections
import csv
import json
import sys

CLASS_OF = {  # kernel-name fragments -> bench.py roofline class
    "cross_attn_lean_kernel": "cross_attn", "self_attn_lean_kernel": "self_attn",
    "proj_ln_tile_kernel<5, 0": "qkv", "proj_ln_tile_kernel<5, 2": "cross_q", "proj_ln_tile_kernel<5, 3": "fc1",
    "proj_res_tile_kernel<5, 5, 4": "fc2", "proj_res_tile_kernel<5, 5, 1": "attn_out",
    "logits_tile_kernel<5": "logits", "enc_flash_attn_kernel": "enc_flash_attn", "logits_rules_kernel": "logits_rules",
    "logits_sumexp_kernel": "logits_sumexp", "logits_stats_kernel": "logits_stats", "logits_sample_kernel": "logits_sample",
}


def per_kernel(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("mw::(anonymous namespace)::", "")
        a = agg[name]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


def main():
    fetch, write, out = sys.argv[1:4]
    f, w = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    res = {}
    for name, (n, kib) in f.items():
        cls = next((c for frag, c in CLASS_OF.items() if frag in name), None)
        if cls is None:
            continue
        wn, wkib = w.get(name, [0, 0.0])
        rd = 2.0 * kib / n * 1024.0
        wr = wkib / wn * 1024.0 if wn else 0.0
        res[cls] = {"kernel": name[:120], "launches": n, "fetch_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                    "hbm_bytes_per_launch": round(rd + wr)}
    json.dump({"unit": "bytes per launch", "correction": "FETCH_SIZE KiB x 1024 x 2 (gfx950 wide reads), WRITE_SIZE KiB x 1024",
               "kernels": res}, open(out, "w"), indent=1)
    for k, v in sorted(res.items()):
        print(k, v["hbm_bytes_per_launch"])


if __name__ == "__main__":
    main()
