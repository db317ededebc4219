#!/bin/bash
# usage: tools/bench_ms.sh LABEL [bench args...]   (env vars select kernels)
label=$1; shift
timeout 300 python bench.py --cpu-frames 0 --steps ${STEPS:-3} --warmup 1 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['ms_per_step'], d['value'], d['roofline']['frac'])"
