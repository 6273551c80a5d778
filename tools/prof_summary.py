#!/usr/bin/env python3
"""Text summary of a tools/prof_round.sh session: python tools/prof_summary.py gpurun_out/prof_r02 > profiles/r02_summary.txt"""
import csv, glob, json, os, re, sys
from collections import defaultdict

O = sys.argv[1]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", n)
    return n if len(n) <= 58 else n[:57] + "~"


def stats(path, title, top=16):
    rows = list(csv.DictReader(open(path)))
    print(f"\n== {title}")
    print(f"{'kernel':58s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_ms':>10s} {'pct':>6s}")
    for r in rows[:top]:
        print(f"{short(r['Name']):58s} {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:10.2f} {float(r['MinNs']) / 1e3:10.2f} "
              f"{float(r['MaxNs']) / 1e3:10.2f} {float(r['TotalDurationNs']) / 1e6:10.3f} {float(r['Percentage']):6.2f}")


print("# rocprofv3 summaries of one profiling session on MI355X (tools/prof_round.sh): kernel trace + stats of")
print("# 'python bench.py --no-cpu-baseline --no-material --no-train --no-image' (fused plan), the same with --plan staged,")
print("# 'python tools/bench_material.py'; PMC passes each in its own run (per-launch means of the dominant kernel in")
print("# profiles/pmc_k_cache_fused.json, the material stage's kernels in profiles/*_material_pmc_counters.txt).")
p = os.path.join(O, "pmc_k_cache_fused.json")
if os.path.exists(p):
    print("# kernel sources:", json.load(open(p))["source_hash"])
stats(os.path.join(O, "fused", "fused_kernel_stats.csv"), "kernel trace, fused plan (default bench; includes the transient line's kernels and the staged separate pass)")
if os.path.exists(os.path.join(O, "fused1", "fused1_kernel_stats.csv")):
    stats(os.path.join(O, "fused1", "fused1_kernel_stats.csv"), "kernel trace, fused plan with ONE wavefront per ray (--plan fused1; rc_set_fused 3: the round-1/2 kernel)", 6)
stats(os.path.join(O, "staged", "staged_kernel_stats.csv"), "kernel trace, launch-per-stage plan (--plan staged; rc_set_fused 0: separate gather / MLP kernels)")
if os.path.exists(os.path.join(O, "material", "material_kernel_stats.csv")):
    stats(os.path.join(O, "material", "material_kernel_stats.csv"), "kernel trace, material stage (tools/bench_material.py: 1024 primary rays, 32 768 secondary rays per step)", 26)
    print(open(os.path.join(O, "bench_material.txt")).read().strip().splitlines()[-1])
acc = defaultdict(list)
for f in glob.glob(os.path.join(O, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_cache_fused" in r["Kernel_Name"] or "k_cache_shader" in r["Kernel_Name"]:
            acc[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
print("\n== PMC (per-launch means)")
for (k, c), v in sorted(acc.items()):
    print(f"{k:58s} {c:26s} launches={len(v):5d} mean={sum(v) / len(v):16.1f}")
