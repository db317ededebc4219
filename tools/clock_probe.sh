#!/bin/bash
# usage: tools/clock_probe.sh LABEL [env assignments...] -- samples sclk / power while bench.py loops
label=$1; shift
env "$@" python bench.py --cpu-frames 0 --steps 150 --warmup 2 > /tmp/probe_$label.json 2>/dev/null &
pid=$!
sleep 9
for i in 1 2 3 4; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Average Graphics Package Power|Current Socket Graphics Package Power" | tr '\n' ' '
  echo
  sleep 0.7
done
wait $pid
python -c "import json; d=json.loads(open('/tmp/probe_$label.json').read().strip().splitlines()[-1]); print('$label', 'ms/step', d['ms_per_step'])"
