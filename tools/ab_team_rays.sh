#!/bin/bash
# VERDICT r3 item 4 experiment: k_cache_fused_team with ONE 8-wave workgroup per CU on one ring (RC_TEAM_RAYS=4; ring chunks of
# 96 or 64 fragments) against the product (two 4-wave workgroups per CU, two rings).  Correctness first (bitwise against the
# one-wave kernel and the staged plan, through the parity tests), then same-box timing, alternating.
for v in rays4_96 rays4_64; do
  echo "== $v: parity tests"
  RC_HIP_LIBRARY=$PWD/tools/diag/$v/librc_hip.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_wave or fused_plan or cache_render_256" 2>&1 | tail -2
done
for i in 1 2; do
  for v in product rays4_96 rays4_64; do
    lib=tools/diag/$v/librc_hip.so; [ $v == product ] && lib=neural-radiance-caching_amd/librc_hip.so
    echo "== round $i $v"
    RC_HIP_LIBRARY=$PWD/$lib timeout -k 10 120 python tools/time_fused.py 1024 600 2>/dev/null | head -1
    RC_HIP_LIBRARY=$PWD/$lib timeout -k 10 120 python tools/time_fused.py 16384 100 2>/dev/null | head -1
  done
done
