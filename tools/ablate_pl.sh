#!/bin/bash
# Ablations of the software-pipelined scoring kernel (ablation build): AASR_DBG bits 64 no transcendentals (v_mov in
# place of v_exp_f32), 128 no close logic, 16 no barriers.  ms per 10^6 frames of configs[1].
export AASR_LIBDIR=$(pwd)/aaltoasr_amd/lib_ablation
for dbg in 0 64 128 192 16 208; do
  AASR_DBG=$dbg tools/bench_ms.sh "pl AASR_DBG=$dbg" --workload gmm --secondary 0
done
