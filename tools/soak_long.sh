#!/bin/bash
# the sweeps of tools/soak.sh at ten times the size (about 20 GPU-minutes); gpurun_out/soak_long/*.log
mkdir -p gpurun_out/soak_long
B=${1:-10000}
for i in 1 2 3 4 5 6; do timeout 1500 python tools/fuzz_parity.py $((B + i)) 500 2>&1 | grep -i "failures\|FAIL\|routed\|probe" | tail -8 >> gpurun_out/soak_long/parity.log; done
timeout 1500 python tools/fuzz_parity.py $((B + 50)) 100 big 2>&1 | grep -i "failures\|FAIL" | tail -5 >> gpurun_out/soak_long/parity_big.log
for i in 1 2 3 4; do AASR_FUZZ_TOL=6e-5 timeout 1500 python tools/fuzz_fullcov.py $((B + 100 + i)) 500 2>&1 | grep "NOTE\|prec=\|failures" >> gpurun_out/soak_long/fullcov.log; done
for i in 1 2; do timeout 1500 python tools/fuzz_features.py $((B + 200 + i)) 500 2>&1 | grep -i "failures\|FAIL" | tail -5 >> gpurun_out/soak_long/features.log; done
timeout 1500 python tools/fuzz_recipe.py $((B + 300)) 500 2>&1 | grep -i "failures\|FAIL" | tail -5 >> gpurun_out/soak_long/recipe.log
timeout 1500 python tools/fuzz_speakers.py $((B + 400)) 200 2>&1 | grep -i "failures\|FAIL" | tail -5 >> gpurun_out/soak_long/speakers.log
timeout 1500 python tools/fuzz_subspace.py $((B + 500)) 300 2>&1 | grep -i "failures\|FAIL\|pcgmm" | tail -5 >> gpurun_out/soak_long/subspace.log
grep -c "failures: 0" gpurun_out/soak_long/*.log
grep -h "FAIL\|failures: [1-9]" gpurun_out/soak_long/*.log | head -20
# models fitted to data (round 6): 120 seeds x 12 models, a third of them scored off the data
bash tools/fuzz_fitted_many.sh $((B + 600)) 120 12 soak_long_$B | tail -3 >> gpurun_out/soak_long/fitted.log
tail -3 gpurun_out/soak_long/fitted.log
