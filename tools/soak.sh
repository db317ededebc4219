mkdir -p gpurun_out/soak
for s in 9101 9102 9103; do timeout 900 python tools/fuzz_parity.py $s 400 2>&1 | grep -i "failures\|FAIL" | tail -5 >> gpurun_out/soak/parity.log; done
timeout 600 python tools/fuzz_parity.py 9201 30 big 2>&1 | grep -i "failures\|FAIL" | tail -5 >> gpurun_out/soak/parity_big.log
for s in 9301 9302; do timeout 600 python tools/fuzz_fullcov.py $s 400 2>&1 | tail -12 >> gpurun_out/soak/fullcov.log; done
timeout 600 python tools/fuzz_features.py 9401 400 2>&1 | grep -i "failures\|FAIL" | tail -5 >> gpurun_out/soak/features.log
timeout 600 python tools/fuzz_recipe.py 9501 200 2>&1 | grep -i "failures\|FAIL" | tail -5 >> gpurun_out/soak/recipe.log
timeout 600 python tools/fuzz_speakers.py 9601 80 2>&1 | grep -i "failures\|FAIL" | tail -5 >> gpurun_out/soak/speakers.log
timeout 600 python tools/fuzz_subspace.py 9701 100 2>&1 | grep -i "failures\|FAIL" | tail -5 >> gpurun_out/soak/subspace.log
timeout 600 python tools/fuzz_wide.py 9801 200 2>&1 | grep -i "failures\|FAIL" | tail -5 >> gpurun_out/soak/wide.log
tail -n 20 gpurun_out/soak/*.log
