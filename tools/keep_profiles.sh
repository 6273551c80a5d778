#!/bin/bash
# Copy what a tools/prof_round.sh session left under gpurun_out/prof_$1 into profiles/ (the tracked copies).
set -e
R=${1:-r02}
O=gpurun_out/prof_$R
cp $O/pmc_k_cache_fused.json profiles/pmc_k_cache_fused.json
cp $O/fused_phase_stamps.json profiles/fused_phase_stamps.json
cp $O/fused_phase_stamps.txt profiles/${R}_fused_phase_stamps.txt
for m in tile strip; do cp $O/fused_phase_stamps_$m.json profiles/fused_phase_stamps_$m.json; cp $O/fused_phase_stamps_$m.txt profiles/${R}_fused_phase_stamps_$m.txt; done
cp $O/fused/fused_kernel_stats.csv profiles/${R}_kernel_stats.csv
cp $O/staged/staged_kernel_stats.csv profiles/${R}_kernel_stats_staged_plan.csv
cp $O/fused1/fused1_kernel_stats.csv profiles/${R}_kernel_stats_one_wave_per_ray.csv
cp $O/material/material_kernel_stats.csv profiles/${R}_material_kernel_stats.csv
cp $O/material_pmc_counters.txt profiles/${R}_material_pmc_counters.txt
[ -f $O/train/train_kernel_stats.csv ] && cp $O/train/train_kernel_stats.csv profiles/${R}_train_backward_kernel_stats.csv
[ -f $O/fused_critical_path.txt ] && cp $O/fused_critical_path.txt profiles/${R}_fused_critical_path.txt
python tools/prof_summary.py $O > profiles/${R}_summary.txt
python -c "
import json, nrc_amd
from nrc_amd import rc_ext
h = rc_ext.source_hash()
for f in ('pmc_k_cache_fused.json', 'fused_phase_stamps.json', 'fused_phase_stamps_tile.json', 'fused_phase_stamps_strip.json'):
    print(f, json.load(open('profiles/' + f))['source_hash'] == h)"
