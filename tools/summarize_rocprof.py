#!/usr/bin/env python3
"""Condense a rocprofv3 *_kernel_stats.csv into a short table (kernel names shortened)."""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(?:<[^(]{0,40}>)?)", name)
    return (m.group(1) if m else name)[:70]


def main(path, limit=12):
    rows = list(csv.DictReader(open(path)))
    print(f"{'kernel':70s} {'calls':>7s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_ms':>10s} {'pct':>6s}")
    for r in rows[:limit]:
        print(f"{short(r['Name']):70s} {int(r['Calls']):7d} {float(r['AverageNs'])/1e3:10.2f} {float(r['MinNs'])/1e3:10.2f} "
              f"{float(r['MaxNs'])/1e3:10.2f} {float(r['TotalDurationNs'])/1e6:10.2f} {float(r['Percentage']):6.2f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 12)
