#!/bin/bash
mkdir -p gpurun_out/r6g
timeout 2400 python -m pytest tests/test_gmm_gpu.py tests/test_pivot_groups_gpu.py tests/test_cluster_gpu.py tests/test_mixed_gpu.py tests/test_fuzz_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > gpurun_out/r6g/pytest.log 2>&1; tail -5 gpurun_out/r6g/pytest.log
timeout 600 bash tools/kstats.sh r6g_fitted_stat -- python tools/bench_fitted.py stationary 5 > gpurun_out/r6g/fitted_stat.log 2>&1; grep "parts\|engine path" gpurun_out/kstats_r6g_fitted_stat/log.txt | cut -c1-600; tail -16 gpurun_out/r6g/fitted_stat.log | head -8
timeout 600 bash tools/kstats.sh r6g_fitted_speech -- python tools/bench_fitted.py speechlike 5 > gpurun_out/r6g/fitted_speech.log 2>&1; grep "parts\|engine path" gpurun_out/kstats_r6g_fitted_speech/log.txt | cut -c1-600; tail -16 gpurun_out/r6g/fitted_speech.log | head -8
bash tools/fuzz_fitted_many.sh 400 12 12 r6g
