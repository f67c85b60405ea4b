#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into per-launch HBM bytes per kernel.

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE counts 128-byte fabric read requests as 64 bytes
for wide (16 B per lane) coalesced reads, so the read side is doubled; WRITE_SIZE is taken as reported.
usage: collect_traffic.py <fetch_csv> <write_csv> <out_json> [workload] [source tag]
"""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    tot, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        m = re.search(r"(sdag_attn_main|sdag_attn_combine|dense_scan|topk_merge_kernel|rescore_kernel)", r["Kernel_Name"])
        if not m:
            continue
        tot[m.group(1)] += float(r["Counter_Value"])
        n[m.group(1)] += 1
    return {k: (tot[k] / n[k], n[k]) for k in tot}


def main(fetch_csv, write_csv, out, mode="cfg3", source=""):
    f = per_kernel(fetch_csv, "FETCH_SIZE")
    w = per_kernel(write_csv, "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) | set(w)):
        fk, nf = f.get(k, (0.0, 0))
        wk, nw = w.get(k, (0.0, 0))
        res[k] = {"launches": max(nf, nw), "fetch_size_kib_raw": fk, "write_size_kib_raw": wk,
                  "read_bytes_corrected": 2 * fk * 1024, "write_bytes": wk * 1024,
                  "hbm_bytes_per_launch": 2 * fk * 1024 + wk * 1024}
    try:
        prev = json.load(open(out))
    except Exception:
        prev = {}
    prev[mode] = res
    if "sdag_attn_main" in res:
        prev["sdag_attn_main_bytes_per_launch_" + mode] = res["sdag_attn_main"]["hbm_bytes_per_launch"]
    if source:
        prev["source"] = f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py on the builder's GPU lease ({source}); not measured in this run"
    res = prev
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:6])
