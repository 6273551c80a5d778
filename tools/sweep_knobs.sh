#!/bin/bash
# Re-tune of the run-time knobs on the current sources (same box, back to back): CUs left to the EnvMap beside the last level of
# the material stage's trace (RC_ENV_RESERVE), priority scheme of the fused kernel (RC_FUSED_PRIO).
for r in 0 40 48 56 64 72 80; do
  RC_ENV_RESERVE=$r python tools/bench_material.py 2>/dev/null | python -c "import sys,ast; d=ast.literal_eval(sys.stdin.read().strip().splitlines()[-1]); print('RC_ENV_RESERVE=$r material ms_per_step', round(d['ms_per_step'],4))"
done
for p in 0 1 2 3 4 5; do
  echo "RC_FUSED_PRIO=$p: $(RC_FUSED_PRIO=$p python tools/time_fused.py 1024 600 2>/dev/null | head -1)"
done
