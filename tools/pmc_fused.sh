#!/bin/bash
# PMC passes of the default bench (two-wavefront fused kernel), each counter group in its own run
set -e
ROOT=$PWD
O=$ROOT/gpurun_out/$1
mkdir -p $O
export TMPDIR=/tmp
B="python $ROOT/bench.py --no-cpu-baseline --no-material --no-train --no-image --no-transient --steps 20 --warmup 5"
cd /tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU"; do
  tag=pmc_$(echo $set | cut -d' ' -f1 | tr 'A-Z' 'a-z')
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/$tag -o p -- $B > /dev/null 2> $O/$tag.err
  echo "$tag done"
done
cd $ROOT
python tools/prof_to_json.py $O $O/pmc.json "k_cache_fused_team<true>" || true
