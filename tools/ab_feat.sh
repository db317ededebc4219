#!/bin/bash
# per-kernel averages of the feature chain for the product library and aaltoasr_amd/lib_prev (same box)
for lib in aaltoasr_amd/lib aaltoasr_amd/lib_prev; do
  AASR_LIBDIR=$lib bash tools/kstats.sh ab -- python tools/stage_split.py 10 2>&1 | grep -E "k_spectral|k_temporal|k_mean_sub" | sed "s|^|$lib |" | cut -c1-60,110-190
done
