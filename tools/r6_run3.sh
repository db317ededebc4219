#!/bin/bash
mkdir -p gpurun_out/r6e
timeout 1500 python bench.py > gpurun_out/r6e/bench_default.json 2> gpurun_out/r6e/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6e/bench_default.json").read().strip().splitlines()[-1])
print("value %.4g ms_per_step %.4f roofline frac %.4f kernel_ms %.4f sclk %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["roofline"].get("sclk_mhz")))
print("stage_ms", d["config"]["stage_ms"])
c=d["config"]
print("ladder", c["precision_ladder"].get("scoring_ms"))
for m in c["precision_routing"]["models"]:
    print("routing share %.2f public ratio %.3f engine ratio %.3f n16 %d" % (m["share"], m["ratio_to_all_f16x2"], m["engine_path_scoring_ratio"], m["states_f16x2"]))
fm=c["fitted_model"]; print("baseline engine path ms", fm["baseline_model_engine_path_ms"])
for m in fm["models"]:
    print(m["audio"], "engine ms %.3f ratio %.3f parts %s max err %.3g / engine layout %.3g build s %s clustered %s" % (m["engine_path_ms"], m["ratio_to_baseline_model"], m["engine_parts"], m["max_abs_dll_vs_oracle_64_frames"], m["engine_layout_vs_oracle_64_frames"]["max_abs_dll"], m["seconds"], m.get("clustered")))
print("clustered", c["clustered"].get("ms_per_pass"), "configs1", c["configs1"].get("ms_per_step"), "configs4", c["configs4"].get("ms_per_step"))
print("recipe", c["recipe_e2e"].get("lnabytes_2"))
PY
timeout 600 bash tools/kstats.sh r6e_fitted_stat -- python tools/bench_fitted.py stationary 5 > gpurun_out/r6e/fitted_stat.log 2>&1; tail -22 gpurun_out/r6e/fitted_stat.log
timeout 600 bash tools/kstats.sh r6e_fitted_speech -- python tools/bench_fitted.py speechlike 5 > gpurun_out/r6e/fitted_speech.log 2>&1; tail -22 gpurun_out/r6e/fitted_speech.log
