#!/bin/bash
# Same-box A/B of the headline line: product library against RC_HIP_LIBRARY=$1, alternating, 3 rounds.
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --no-transient --no-material --no-train 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('product', round(d['value']), d['roofline']['avg_launch_ms'])"
  RC_HIP_LIBRARY=$1 python bench.py --no-cpu-baseline --no-transient --no-material --no-train 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('other  ', round(d['value']), d['roofline']['avg_launch_ms'])"
done
