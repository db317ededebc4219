mkdir -p gpurun_out/soak
for s in ${SOAK_SEEDS:-9101 9102 9103}; do timeout 900 python tools/fuzz_parity.py $s 400 2>&1 | grep -i "failures\|FAIL\|routed\|probe" | tail -8 >> gpurun_out/soak/parity.log; done
timeout 600 python tools/fuzz_parity.py ${SOAK_BASE:-9}201 30 big 2>&1 | grep -i "failures\|FAIL" | tail -5 >> gpurun_out/soak/parity_big.log
for s in ${SOAK_BASE:-9}301 ${SOAK_BASE:-9}302; do timeout 600 python tools/fuzz_fullcov.py $s 400 2>&1 | tail -12 >> gpurun_out/soak/fullcov.log; done
timeout 600 python tools/fuzz_features.py ${SOAK_BASE:-9}401 400 2>&1 | grep -i "failures\|FAIL" | tail -5 >> gpurun_out/soak/features.log
timeout 600 python tools/fuzz_recipe.py ${SOAK_BASE:-9}501 200 2>&1 | grep -i "failures\|FAIL" | tail -5 >> gpurun_out/soak/recipe.log
timeout 600 python tools/fuzz_speakers.py ${SOAK_BASE:-9}601 80 2>&1 | grep -i "failures\|FAIL" | tail -5 >> gpurun_out/soak/speakers.log
timeout 600 python tools/fuzz_subspace.py ${SOAK_BASE:-9}701 100 2>&1 | grep -i "failures\|FAIL" | tail -5 >> gpurun_out/soak/subspace.log
timeout 600 python tools/fuzz_wide.py ${SOAK_BASE:-9}801 200 2>&1 | grep -i "failures\|FAIL" | tail -5 >> gpurun_out/soak/wide.log
# models fitted to data, a third of them scored off the data (round 6: part of the soak; flat 1e-4 on every visible value)
bash tools/fuzz_fitted_many.sh ${SOAK_BASE:-9}901 ${SOAK_FITTED_SEEDS:-46} 12 soak_${SOAK_BASE:-9} | tail -4 >> gpurun_out/soak/fitted.log
tail -n 20 gpurun_out/soak/*.log
