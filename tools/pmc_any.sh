#!/bin/bash
# usage: tools/pmc_any.sh "COUNTER ..." KERNEL_SUBSTRING -- CMD...   (one rocprofv3 --pmc pass, mean per launch)
ctr=$1; kern=$2; shift 3
root=$(pwd); out=$root/gpurun_out/pmc_any; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctr --output-format csv -d $out -- bash -c "cd $root && $*" > $out/log.txt 2>&1
cd $root
python - "$out" "$kern" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
for p in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for c, d in sorted(acc.items()):
    v = list(d.values()); print("%-34s %.4g per launch (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
