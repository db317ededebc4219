#!/bin/bash
# the multi-pivot launch's row cuts on the fitted stationary model (ablation build): forced cut counts / assumed overheads
export AASR_LIBDIR=$(pwd)/aaltoasr_amd/lib_ablation
for c in 1 3 6 10 16; do
  AASR_CUT_OVERHEAD=$c timeout 300 python tools/bench_fitted.py stationary 5 2>/dev/null | grep "engine path" | tr '\n' ' '; echo " [overhead $c]"
done
for r in 8 9 10 12 14 16 20 24 32 40; do
  AASR_SPLITS=$r timeout 300 python tools/bench_fitted.py stationary 5 2>/dev/null | grep "engine path" | tr '\n' ' '; echo " [splits $r]"
done
AASR_TWO_LEVEL=0 timeout 300 python tools/bench_fitted.py stationary 5 2>/dev/null | grep "engine path\|^parts" | cut -c1-300 | tr '\n' ' '; echo " [two-level off]"
