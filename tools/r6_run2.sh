#!/bin/bash
# round 6, GPU run 2: the slab-constant layout -- calibration sweep (every state forced into one SC part) next to the plain
# layout's; the pivot-group tests on the product build
mkdir -p gpurun_out/r6b
AASR_EXP_FORCE_SC=1 AASR_LIBDIR=$(pwd)/aaltoasr_amd/lib_open timeout 1500 python tools/exp_calib.py 200 16 8 sc > gpurun_out/r6b/calib_sc.log 2>&1; tail -4 gpurun_out/r6b/calib_sc.log
timeout 1500 python -m pytest tests/test_gmm_gpu.py tests/test_pivot_groups_gpu.py tests/test_mixed_gpu.py tests/test_cluster_gpu.py tests/test_pipeline_gpu.py -q -m gpu > gpurun_out/r6b/pytest.log 2>&1
tail -15 gpurun_out/r6b/pytest.log
