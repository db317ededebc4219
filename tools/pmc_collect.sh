#!/bin/bash
# usage: tools/pmc_collect.sh OUT.json [bench args...]
# rocprofv3 PMC passes (one counter group per run, as MI355X_MICROARCH.md prescribes) over
# `bench.py --cpu-frames 0`, summarised per launch of the dominant scoring kernel.
out=$1; shift
root=$(pwd)
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf $root/gpurun_out/pmc/p$i
  rocprofv3 --pmc $grp --output-format csv -d $root/gpurun_out/pmc/p$i -- python $root/bench.py --workload gmm --cpu-frames 0 --secondary 0 --steps 3 --warmup 1 "$@" > /dev/null 2>&1
done
cd $root
python tools/pmc_summarize.py gpurun_out/pmc "$out"
