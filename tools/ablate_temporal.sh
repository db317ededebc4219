#!/bin/bash
# phase ablations of k_temporal_fused (AASR_TEMP_DBG: 1 no transform products, 2 no second difference/normalisation,
# 4 no first difference, 8 no matrix staging) and k_mean_subtract_tiled (AASR_CMS_DBG: 1 no window sums, 2 no block
# sums, 4 no staging) in the ablation build: average kernel duration per setting
for d in 0 1 2 4 8 15; do
  AASR_TEMP_DBG=$d AASR_LIBDIR=aaltoasr_amd/lib_ablation bash tools/kstats.sh abl -- python tools/stage_split.py 10 2>&1 | grep "k_temporal_fused" | sed "s/^/temporal dbg=$d /" | cut -c1-50,100-170
done
for d in 0 1 2 4 7; do
  AASR_CMS_DBG=$d AASR_LIBDIR=aaltoasr_amd/lib_ablation bash tools/kstats.sh abl -- python tools/stage_split.py 10 2>&1 | grep "k_mean_subtract_tiled" | sed "s/^/cms dbg=$d /" | cut -c1-50,100-170
done
