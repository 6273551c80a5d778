#!/bin/bash
# Table-gradient scatter, level by level (timing experiment; run on the GPU box): tools/bench_train.py with every level but one
# skipped (RC_SCATTER_SKIP), then LDS accumulation of the 32^3 F = 1 level (RC_SCATTER_LDS_MAX=32768) with 16 ... 256 workgroups.
echo "== all levels (product)"; python tools/bench_train.py 2>/dev/null
for l in 0 1 2 3 4 5 6 7; do
  m=$(( 255 & ~(1 << l) ))
  echo "== only level $l (mask $m)"; RC_SCATTER_SKIP=$m python tools/bench_train.py 2>/dev/null
done
echo "== no table gradients at all"; RC_SCATTER_SKIP=255 python tools/bench_train.py 2>/dev/null
for w in 16 32 64 128 256; do
  echo "== 32^3 F=1 level in LDS, $w workgroups"; RC_SCATTER_LDS_MAX=32768 RC_SCATTER_BIG_WGS=$w python tools/bench_train.py 2>/dev/null
done
