# L1 / L2 counters of the material stage's kernels (k_level above all): separate passes, kernel-trace only.
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_level; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/$tag -o p -- python $GRAFT_REPO_ROOT/tools/bench_material.py > $O/$tag.log 2>&1 || echo "FAILED $set"
done
ls -R $O | head -50
