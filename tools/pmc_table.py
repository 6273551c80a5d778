#!/usr/bin/env python3
"""Per-kernel table of rocprofv3 PMC passes: python tools/pmc_table.py <prof dir> "<glob of pass dirs>"
For every kernel the longest dispatch of each pass (the full-size launch) with its duration."""
import csv, glob, os, re, sys
from collections import defaultdict

prof, pat = sys.argv[1], sys.argv[2]
tab = defaultdict(dict)
for p in sorted(glob.glob(os.path.join(prof, pat, "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(p)):
        m = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
        if not m:
            continue
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        cur = tab[m.group(1)].get(r["Counter_Name"])
        if cur is None or dur > cur[0]:
            tab[m.group(1)][r["Counter_Name"]] = (dur, float(r["Counter_Value"]))
print("# counter values of the longest dispatch of each kernel (one rocprofv3 --pmc pass per counter group)")
for k in sorted(tab, key=lambda k: -max(v[0] for v in tab[k].values())):
    dur = max(v[0] for v in tab[k].values())
    if dur < 10000:
        continue
    print(f"{k}   ({dur / 1e3:.0f} us)")
    for c, (t, v) in sorted(tab[k].items()):
        print(f"    {c:36s} {v:14.5g}")
