#!/bin/bash
# Same-box A/B of the F = 4 density cell records (RC_REC4_LEVELS = 0 | 1 | 2 builds; the product is 2): material stage, 2 rounds.
for i in 1 2 3; do
  for n in 0 1 2; do
    lib=tools/diag/rec4_$n/librc_hip.so
    [ $n == 2 ] && lib=neural-radiance-caching_amd/librc_hip.so
    RC_HIP_LIBRARY=$PWD/$lib python tools/bench_material.py 2>/dev/null | python -c "import sys,ast; d=ast.literal_eval(sys.stdin.read().strip().splitlines()[-1]); print('round $i RC_REC4_LEVELS=$n material ms_per_step', round(d['ms_per_step'],4))"
  done
done
