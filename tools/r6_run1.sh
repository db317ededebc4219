#!/bin/bash
# round 6, GPU run 1: the rewritten centred kernel (tests + timing) and the calibration sweep on the open-limits build
mkdir -p gpurun_out/r6a
timeout 1500 python -m pytest tests/test_gmm_gpu.py tests/test_pivot_groups_gpu.py tests/test_cluster_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu > gpurun_out/r6a/pytest.log 2>&1
tail -5 gpurun_out/r6a/pytest.log
timeout 300 python tools/bench_centred.py > gpurun_out/r6a/bench_centred.log 2>&1; tail -2 gpurun_out/r6a/bench_centred.log
timeout 600 bash tools/kstats.sh r6a_fitted_stat -- python tools/bench_fitted.py stationary 5 > gpurun_out/r6a/fitted_stat.log 2>&1; tail -25 gpurun_out/r6a/fitted_stat.log
AASR_LIBDIR=$(pwd)/aaltoasr_amd/lib_open timeout 1500 python tools/exp_calib.py 200 16 8 a > gpurun_out/r6a/calib.log 2>&1; tail -4 gpurun_out/r6a/calib.log
