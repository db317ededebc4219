#!/bin/bash
# per-kernel averages of the feature chain for the product library and experiment builds (same box):
#   tools/ab_feat.sh [libdir ...]      default: aaltoasr_amd/lib aaltoasr_amd/lib_prev
libs="$@"; [ -z "$libs" ] && libs="aaltoasr_amd/lib aaltoasr_amd/lib_prev"
for lib in $libs; do
  AASR_LIBDIR=$lib bash tools/kstats.sh ab -- python tools/stage_split.py 10 2>&1 | grep -E "k_spectral|k_temporal|k_mean_sub" | sed "s|^|$lib |" | cut -c1-60,110-190
done
