#!/bin/bash
# Effective clock of the scoring kernel under each ablation: GRBM_GUI_ACTIVE (cycles the GPU was busy, at whatever
# clock the power cap allowed) per launch from a rocprofv3 --pmc pass, next to the ms per launch of a plain run.
# usage: tools/ablate_clock.sh [f16x2|bf16x3] "0 1 3 259"
prec=${1:-f16x2}; list=${2:-"0 1 3 259"}
export AASR_LIBDIR=$(pwd)/aaltoasr_amd/lib_ablation
for dbg in $list; do
  export AASR_DBG=$dbg
  echo "== $prec AASR_DBG=$dbg"
  tools/bench_ms.sh "ms" --workload gmm --secondary 0 --precision $prec
  tools/pmc_any.sh "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" k_gmm_diag_score_bf16x3 -- python bench.py --workload gmm --secondary 0 --cpu-frames 0 --steps 2 --warmup 1 --precision $prec
done
