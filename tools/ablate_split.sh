#!/bin/bash
# Ablations of the split-operand scoring kernel (ablation build, aaltoasr_amd/lib_ablation): ms per 10^6 frames of
# configs[1] with parts of the kernel removed.  usage: tools/ablate_split.sh [f16x2|bf16x3]
prec=${1:-f16x2}
export AASR_LIBDIR=$(pwd)/aaltoasr_amd/lib_ablation
for dbg in 0 1 257 3 259 17; do
  AASR_DBG=$dbg tools/bench_ms.sh "$prec AASR_DBG=$dbg" --workload gmm --secondary 0 --precision $prec
done
