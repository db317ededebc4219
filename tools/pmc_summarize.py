"""Summarise rocprofv3 --pmc CSVs (tools/pmc_collect.sh) into one JSON: mean counter value per
launch of the dominant k_gmm_* scoring kernel (the kernel with the largest launch count x grid)."""
import csv, glob, json, os, sys
from collections import defaultdict

root, out = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> values (summed over dims per dispatch)
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    per_dispatch = defaultdict(float)
    names = {}
    for row in csv.DictReader(open(path)):
        k = row.get("Kernel_Name", "")
        if "k_gmm" not in k:
            continue
        key = (row.get("Dispatch_Id"), row.get("Counter_Name"))
        per_dispatch[key] += float(row.get("Counter_Value", 0))
        names[key] = k
    for (did, cname), v in per_dispatch.items():
        acc[names[(did, cname)]][cname].append(v)
if not acc:
    sys.exit("no k_gmm kernel in the counter files")
kernel = max(acc, key=lambda k: sum(len(v) for v in acc[k].values()))
res = {"frames_per_launch": 1000000, "kernel": kernel.split("(")[0]}
for cname, vals in sorted(acc[kernel].items()):
    res[cname] = {"mean_per_launch": sum(vals) / len(vals), "launches": len(vals)}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res)[:600])
