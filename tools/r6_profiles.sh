#!/bin/bash
# round 6: the driver-shaped bench line + the kernel traces / counters that go to profiles/ (tag = $1)
tag=${1:-a}
mkdir -p gpurun_out/r6p
timeout 1500 python bench.py > gpurun_out/r6p/bench_default_$tag.json 2> gpurun_out/r6p/bench_default_$tag.err; tail -c 600 gpurun_out/r6p/bench_default_$tag.json | head -c 300; echo
timeout 900 bash tools/kstats.sh r6_full_chain_f16x2 -- python bench.py --cpu-frames 0 --secondary 0 --steps 20 --warmup 3 > gpurun_out/r6p/kstats_full.log 2>&1; tail -8 gpurun_out/r6p/kstats_full.log
timeout 900 bash tools/kstats.sh r6_gmm_f16x2 -- python bench.py --workload gmm --cpu-frames 0 --secondary 0 --steps 10 --warmup 2 > gpurun_out/r6p/kstats_gmm.log 2>&1; tail -4 gpurun_out/r6p/kstats_gmm.log
timeout 900 bash tools/kstats.sh r6_cluster -- python tools/bench_cluster.py > gpurun_out/r6p/kstats_cluster.log 2>&1; tail -9 gpurun_out/r6p/kstats_cluster.log
timeout 600 bash tools/kstats.sh r6_fitted_stationary_engine_path -- python tools/bench_fitted.py stationary 5 > gpurun_out/r6p/fitted_stat.log 2>&1; tail -9 gpurun_out/r6p/fitted_stat.log
timeout 600 bash tools/kstats.sh r6_fitted_speechlike_engine_path -- python tools/bench_fitted.py speechlike 5 > gpurun_out/r6p/fitted_speech.log 2>&1; tail -9 gpurun_out/r6p/fitted_speech.log
timeout 900 bash tools/pmc_collect.sh gpurun_out/r6p/r6_gmm_f16x2_pmc_aligned.json --out-pitch aligned > gpurun_out/r6p/pmc.log 2>&1; tail -3 gpurun_out/r6p/pmc.log
