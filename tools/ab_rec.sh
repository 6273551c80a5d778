#!/bin/bash
# Same-box A/B of the hashed-level cell records (RC_REC_LEVELS = 0 | 1 | 2 builds under tools/diag/rec<n>/; the product is 2):
# the fused kernel per launch (tools/time_fused.py) and the material stage (tools/bench_material.py), alternating, 2 rounds.
for i in 1 2; do
  for n in 0 1 2; do
    lib=tools/diag/rec$n/librc_hip.so
    [ $n == 2 ] && lib=neural-radiance-caching_amd/librc_hip.so
    echo "== round $i, RC_REC_LEVELS=$n"
    RC_HIP_LIBRARY=$PWD/$lib python tools/time_fused.py 1024 600 2>/dev/null | head -1
    RC_HIP_LIBRARY=$PWD/$lib python tools/time_fused.py 16384 100 2>/dev/null | head -1
    RC_HIP_LIBRARY=$PWD/$lib python tools/bench_material.py 2>/dev/null | python -c "import sys,ast; d=ast.literal_eval(sys.stdin.read().strip().splitlines()[-1]); print('material ms_per_step', round(d['ms_per_step'],4))"
  done
done
