#!/bin/bash
python -m pytest tests/test_train.py -m gpu -x -q 2>&1 | tail -2
for w in 256 128 64 32; do echo "== big-table workgroups $w"; RC_SCATTER_BIG_WGS=$w python tools/bench_train.py 2>/dev/null; done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4_train_trace2 -o t -- python $GRAFT_REPO_ROOT/tools/bench_train.py > /dev/null 2>&1
