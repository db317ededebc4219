for d in 0 2 4 8 16 32 64 128 254; do
  rm -rf gpurun_out/kstats_abl; AASR_SPEC_DBG=$d AASR_LIBDIR=aaltoasr_amd/lib_ablation bash tools/kstats.sh abl -- python tools/stage_split.py 10 2>&1 | grep "k_spectral_fused" | sed "s/^/dbg=$d /" | cut -c1-40,100-160
done
