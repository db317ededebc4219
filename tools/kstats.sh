#!/bin/bash
# usage: tools/kstats.sh NAME -- CMD...   rocprofv3 --kernel-trace --stats of CMD; the per-kernel summary
# lands in gpurun_out/kstats_NAME.csv (copy what should be judged to profiles/)
name=$1; shift 2
root=$(pwd); out=$root/gpurun_out/kstats_$name; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- bash -c "cd $root && $*" > $out/log.txt 2>&1
cd $root
# the command may start helper processes with kernels of their own (bench.py's matrix-pipe probe): take the
# summary of the process that ran the engine's kernels
f=$(grep -l "aasr::" $(find $out -name "*kernel_stats.csv") | head -1)
cp "$f" $root/gpurun_out/kstats_$name.csv
python - "$root/gpurun_out/kstats_$name.csv" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:18]:
    print("%-90s calls %5s avg %10.1f us  total %8.2f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
