#!/bin/bash
python -m pytest tests/test_train.py -m gpu -x -q 2>&1 | tail -2
echo "== sliced (product)"; python tools/bench_train.py 2>/dev/null
echo "== RC_SCATTER_SLICED=0"; RC_SCATTER_SLICED=0 python tools/bench_train.py 2>/dev/null
echo "== sliced again"; python tools/bench_train.py 2>/dev/null
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4_train_trace3 -o t -- python $GRAFT_REPO_ROOT/tools/bench_train.py > /dev/null 2>&1
