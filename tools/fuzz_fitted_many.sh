#!/bin/bash
# usage: tools/fuzz_fitted_many.sh FIRST_SEED N_SEEDS [MODELS_PER_SEED] [TAG]  -- tools/fuzz_fitted.py over a range of seeds;
# FAIL lines and every seed's worst figures land in gpurun_out/fuzz_fitted_<TAG>.log
first=${1:-300}; n=${2:-46}; per=${3:-12}; tag=${4:-a}
mkdir -p gpurun_out; log=gpurun_out/fuzz_fitted_$tag.log; : > $log
for ((s=first; s<first+n; s++)); do
  timeout 900 python tools/fuzz_fitted.py $s $per 2>&1 | grep -i "^worst\|^failures\|^FAIL\|Error\|Traceback" | cut -c1-700 >> $log
done
echo "seeds $first..$((first+n-1)), $per models each: $(grep -c '^FAIL' $log) FAIL lines, $(grep -c '^failures: 0' $log) clean seeds of $n"
grep "^FAIL" $log | head -20
python - "$log" <<'PY'
import re, sys, ast
worst = {}
for line in open(sys.argv[1]):
    if line.startswith("worst:"):
        try:
            d = ast.literal_eval(line[len("worst:"):].strip())
        except Exception:
            continue
        for k, v in d.items():
            worst[k] = (worst.get(k, 0) + v) if k in ("n parts",) else max(worst.get(k, 0), v)
print("over all seeds:", worst)
PY
