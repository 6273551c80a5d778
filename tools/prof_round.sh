#!/bin/bash
# Profiling session of a round (run on the GPU box through gpurun, from the repo root): kernel traces of the default
# bench (fused plan), of the staged plan and of the material stage, PMC passes (each in its own run, kernel-trace only),
# the in-kernel phase stamps of the fused kernel (needs `make -C neural-radiance-caching_amd/csrc diag` beforehand; with
# `make ... diag DIAG_EXTRA=-DRC_GATHER_FAKE=0 DIAG_DIR=fake` as well, the per-phase critical-path table).
# Everything lands under gpurun_out/prof_$1; copy what is to be kept into profiles/.
set -e
R=${1:-r02}
ROOT=$PWD
O=$ROOT/gpurun_out/prof_$R
mkdir -p $O
export TMPDIR=/tmp
B="python $ROOT/bench.py --no-cpu-baseline --no-material --no-train --no-image"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fused -o fused -- $B > $O/bench_fused.json 2> $O/fused.err
echo "fused trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fused1 -o fused1 -- $B --no-transient --plan fused1 > $O/bench_fused1.json 2> $O/fused1.err
echo "fused1 (one wavefront per ray) trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/staged -o staged -- $B --no-transient --plan staged > $O/bench_staged.json 2> $O/staged.err
echo "staged trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/material -o material -- python $ROOT/tools/bench_material.py > $O/bench_material.txt 2> $O/material.err
echo "material trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o train -- python $ROOT/tools/bench_train.py > $O/bench_train.txt 2> $O/train.err
echo "train-backward trace done"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU"; do
  tag=pmc_$(echo $set | cut -d' ' -f1 | tr 'A-Z' 'a-z')
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/$tag -o p -- $B --no-transient --steps 20 --warmup 5 > /dev/null 2> $O/$tag.err
  echo "$tag done"
done
# L1 / L2 / fabric counters of the material stage's kernels (the level kernels above all)
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  tag=mat_$(echo $set | cut -d' ' -f1 | tr 'A-Z' 'a-z')
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/$tag -o p -- python $ROOT/tools/bench_material.py > /dev/null 2> $O/$tag.err
  echo "$tag done"
done
cd $ROOT
python tools/prof_to_json.py $O $O/pmc_k_cache_fused.json || echo "prof_to_json failed"
python tools/pmc_table.py $O "mat_*" > $O/material_pmc_counters.txt || echo "pmc_table failed"
if [ -f tools/diag/librc_hip.so ]; then
  python tools/gpu_stamps_fused.py 1024 "" $O/fused_phase_stamps.json > $O/fused_phase_stamps.txt
  RC_STAMP_RAYS=tile python tools/gpu_stamps_fused.py 1024 "" $O/fused_phase_stamps_tile.json > $O/fused_phase_stamps_tile.txt
  RC_STAMP_RAYS=strip python tools/gpu_stamps_fused.py 1024 "" $O/fused_phase_stamps_strip.json > $O/fused_phase_stamps_strip.txt
  echo "stamps done"
fi
if [ -f tools/diag/librc_hip.so ] && [ -f tools/diag/fake/librc_hip.so ]; then
  # per-phase / per-wave critical path of the fused kernel: stamped build against the stamped build without memory time
  python tools/fused_critical_path.py > $O/fused_critical_path.txt
  echo "critical path done"
fi
find $O -name "*.csv" | sort
