#!/bin/bash
# Stand-alone k_hashgrid_fwd times (launch-per-stage plan, per-stage HIP events of each tree's own bench.py) for the trees
# built under tools/diag/bisect/<sha>/ and for the working tree, back to back on one box, twice (order effects).
# Run on the GPU box from the repo root: bash tools/bisect_hashgrid.sh > gpurun_out/bisect_hashgrid.txt
R=$PWD
for pass in 1 2; do
  for d in $R/tools/diag/bisect/* $R; do
    [ -f $d/bench.py ] || continue
    name=$(basename $d)
    [ "$d" == "$R" ] && name=HEAD
    (cd $d && python bench.py --plan staged --no-cpu-baseline --no-material --no-train --no-image --no-transient --steps 100 2> /dev/null |
      python -c "
import json, sys
r = json.loads(sys.stdin.readline())
s = r['stage_ms_separate_pass_staged_plan']
g = {k: round(s[k] * 1e3, 2) for k in ('grid0', 'grid1', 'grid2', 'grid_app')}
print('pass $pass  %-8s' % '$name', g, 'sum %.1f us' % sum(g.values()), ' value %.0f rays/s' % r['value'])
")
  done
done
