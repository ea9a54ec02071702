#!/bin/bash
# usage: scripts/gpu_kstat.sh <tag> <pattern> <bench args...>: rocprofv3 kernel stats of bench.py, rows matching pattern
tag=$1; pat=$2; shift; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=/tmp/prof_$tag
rm -rf $out; mkdir -p $out $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $root/bench.py "$@" > $root/gpurun_out/kstat_$tag.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp "$f" $root/gpurun_out/kstat_${tag}.csv
python3 - "$f" "$pat" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
        print("%-60s calls %4s avg %10.1f us  min %10.1f max %10.1f" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
