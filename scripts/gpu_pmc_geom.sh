#!/bin/bash
# usage: scripts/gpu_pmc_geom.sh <tag> "<counters>" [points] [overlap]
# one rocprofv3 --pmc pass (own run, kernel-trace only) over the geometry builds of scripts/prof_geom.py;
# per-kernel sums over all dispatches land in gpurun_out/pmc_<tag>.csv
tag=$1; shift; ctrs=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=/tmp/pmc_$tag
rm -rf $out; mkdir -p $out $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -- python $root/scripts/prof_geom.py ${1:-10000000} ${2:-0} > $root/gpurun_out/pmc_$tag.log 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
python3 - "$f" $root/gpurun_out/pmc_${tag}.csv <<'PY'
import csv, sys, collections
rows = csv.DictReader(open(sys.argv[1]))
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if "at::native" in n or "rocclr" in n or "rocprim" in n:
        continue
    d = agg.setdefault(n, collections.OrderedDict(calls=set()))
    d["calls"].add(r["Dispatch_Id"])
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
names = sorted({k for d in agg.values() for k in d if k != "calls"})
with open(sys.argv[2], "w") as f:
    f.write("kernel|dispatches|" + "|".join(names) + "\n")
    for n, d in agg.items():
        f.write(n + "|%d|" % len(d["calls"]) + "|".join("%.6g" % d.get(k, 0) for k in names) + "\n")
PY
