#!/bin/bash
mkdir -p gpurun_out/r6f
timeout 1500 python -m pytest tests/test_gmm_gpu.py tests/test_pivot_groups_gpu.py tests/test_cluster_gpu.py tests/test_mixed_gpu.py tests/test_pipeline_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu > gpurun_out/r6f/pytest.log 2>&1; tail -5 gpurun_out/r6f/pytest.log
timeout 300 python tools/bench_centred.py 2>&1 | tail -1
timeout 600 bash tools/kstats.sh r6f_fitted_stat -- python tools/bench_fitted.py stationary 5 > gpurun_out/r6f/fitted_stat.log 2>&1; grep "parts\|engine path" gpurun_out/kstats_r6f_fitted_stat/log.txt | cut -c1-900; tail -16 gpurun_out/r6f/fitted_stat.log | head -9
timeout 600 bash tools/kstats.sh r6f_fitted_speech -- python tools/bench_fitted.py speechlike 5 > gpurun_out/r6f/fitted_speech.log 2>&1; grep "parts\|engine path\|\[two" gpurun_out/kstats_r6f_fitted_speech/log.txt | cut -c1-1500; tail -16 gpurun_out/r6f/fitted_speech.log | head -9
