#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into the per-kernel tables kept under profiles/.

    python tools/prof_summarize.py stats  <*_kernel_stats.csv>
    python tools/prof_summarize.py pmc    <*_counter_collection.csv> [...]
"""
import csv
import sys
from collections import defaultdict


def short(name, width=56):
    name = name.replace("(anonymous namespace)::", "")
    return (name[: width - 1] + "~") if len(name) > width else name


def stats(path):
    rows = list(csv.DictReader(open(path)))
    print(f"{'kernel':56s} {'calls':>7s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_ms':>10s} {'pct':>6s}")
    for r in rows:
        print(f"{short(r['Name']):56s} {int(r['Calls']):7d} {float(r['AverageNs']) / 1e3:10.2f} {float(r['MinNs']) / 1e3:10.2f} "
              f"{float(r['MaxNs']) / 1e3:10.2f} {float(r['TotalDurationNs']) / 1e6:10.3f} {float(r['Percentage']):6.2f}")


def pmc(paths):
    acc = defaultdict(list)
    for p in paths:
        for r in csv.DictReader(open(p)):
            acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        if k.startswith("__amd") or "at::native" in k or "elementwise" in k:
            continue
        print(f"{short(k):56s} {c:26s} launches={len(v):5d} mean={sum(v) / len(v):16.1f}")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2:])
