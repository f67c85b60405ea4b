#!/usr/bin/env python3
"""Per-kernel averages of the counters in rocprofv3 --pmc CSV outputs. usage: pmc_summary.py <dir>..."""
import collections, csv, glob, re, sys
acc = collections.defaultdict(lambda: [0.0, 0])
for d in sys.argv[1:]:
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            m = re.search(r"(sdag_attn_main|sdag_attn_combine|dense_scan|topk_merge_kernel|rescore_kernel)", r["Kernel_Name"])
            if not m:
                continue
            a = acc[(m.group(1), r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (t, n) in sorted(acc.items()):
    print(f"{k:22s} {c:32s} {t / n:16.1f}  (n={n})")
