#!/usr/bin/env python3
"""rocprofv3 counter CSVs of tools/prof_round.sh -> the small JSON files bench.py reads from profiles/
(pmc_k_cache_fused.json): per-launch means of the dominant kernel, tagged with the kernel-source hash they were
measured on (bench.py reports `traffic` only when that hash matches the sources it runs).

    python tools/prof_to_json.py <prof dir> <out json> [kernel substring]
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nrc_amd  # noqa: E402,F401
from nrc_amd import rc_ext  # noqa: E402


def main():
    prof, out = sys.argv[1], sys.argv[2]
    # the dominant kernel of the default plan: the one-wave-per-ray kernel in a split-MFMA build, the two-wave kernel otherwise
    default = "k_cache_fused<true" if rc_ext.mlp_arithmetic() == "bf16x3-split" else "k_cache_fused_team<true>"
    kernel = sys.argv[3] if len(sys.argv) > 3 else default
    acc = defaultdict(list)
    for p in glob.glob(os.path.join(prof, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            if kernel in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not acc:
        raise SystemExit(f"no counter rows for {kernel} under {prof}")
    d = {"source_hash": rc_ext.source_hash(), "kernel": kernel, "launches": {k: len(v) for k, v in acc.items()},
         "command": "rocprofv3 --pmc <counter> --kernel-trace -- python bench.py --no-cpu-baseline --no-transient --no-material "
                    "--no-train --no-image --steps 20 --warmup 5 (one run per counter group, tools/prof_round.sh)"}
    for k, v in acc.items():
        d[k + ("_KiB" if k in ("FETCH_SIZE", "WRITE_SIZE") else "")] = sum(v) / len(v)
    json.dump(d, open(out, "w"), indent=1)
    print(json.dumps(d))


if __name__ == "__main__":
    main()
