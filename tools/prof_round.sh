#!/bin/bash
# Profiling session of a round (run on the GPU box through gpurun): kernel trace of the default bench
# (fused plan), of the staged plan, PMC passes (each in its own run), everything under gpurun_out/prof_$1.
set -e
R=${1:-r02}
O=gpurun_out/prof_$R
mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fused -o fused -- python bench.py --no-cpu-baseline --no-material --no-train --no-image > $O/bench_fused.json 2> $O/fused.err
echo "fused trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/staged -o staged -- python bench.py --no-cpu-baseline --no-transient --no-material --no-train --no-image --plan staged > $O/bench_staged.json 2> $O/staged.err
echo "staged trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o fetch -- python bench.py --no-cpu-baseline --no-transient --no-material --no-train --no-image --steps 20 --warmup 5 > /dev/null 2> $O/pmc_fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o write -- python bench.py --no-cpu-baseline --no-transient --no-material --no-train --no-image --steps 20 --warmup 5 > /dev/null 2> $O/pmc_write.err
echo "write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -o mfma -- python bench.py --no-cpu-baseline --no-transient --no-material --no-train --no-image --steps 20 --warmup 5 > /dev/null 2> $O/pmc_mfma.err
echo "mfma done"
python tools/prof_to_json.py $O $O/pmc_k_cache_fused.json
find $O -name "*.csv" | sort
