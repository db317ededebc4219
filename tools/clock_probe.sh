#!/bin/bash
# usage: tools/clock_probe.sh LABEL [env assignments...] -- samples sclk / power while bench.py loops
label=$1; shift
env "$@" python bench.py --cpu-frames 0 --steps ${PROBE_STEPS:-300} --warmup 2 > /tmp/probe_$label.json 2>/dev/null &
pid=$!
# sample until the run ends; keep the samples taken under load
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Current Socket Graphics Package Power" | tr '\n' ' ' | \
    sed -E 's/.*\(([0-9]+)Mhz\).*Power \(W\): ([0-9.]+).*/\1 MHz \2 W/'
  echo
  sleep 0.4
done | awk '$3+0 > 700' | tail -6
wait $pid
python -c "import json; d=json.loads(open('/tmp/probe_$label.json').read().strip().splitlines()[-1]); print('$label', 'ms/step', d['ms_per_step'])"
