#!/bin/bash
# usage: scripts/gpu_profile.sh <tag> <bench args...>
# runs rocprofv3 --kernel-trace --stats on bench.py and keeps only the small summary files
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=/tmp/prof_$tag
rm -rf $out; mkdir -p $out $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $root/bench.py "$@" > $root/gpurun_out/prof_$tag.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $root/gpurun_out/prof_${tag}_kernel_stats.csv; fi
tail -2 $root/gpurun_out/prof_$tag.log | cut -c1-1500
head -40 $root/gpurun_out/prof_${tag}_kernel_stats.csv | cut -c1-200
t=$(find $out -name "*kernel_trace.csv" | head -1)
if [ -n "$t" ]; then python3 - "$t" $root/gpurun_out/prof_${tag}_sconv_trace.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if "k_sconv" in r["Kernel_Name"] or "k_cconv" in r["Kernel_Name"] or "k_radius" in r["Kernel_Name"]]
keep = keep[-(len(keep) // 3):] if len(keep) > 90 else keep   # last forward only
with open(sys.argv[2], "w") as f:
    f.write("kernel,grid_x,grid_y,duration_us,lds,vgpr\n")
    for r in keep:
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        f.write("%s,%s,%s,%.1f,%s,%s\n" % (n, r["Grid_Size_X"], r["Grid_Size_Y"],
                (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("LDS_Block_Size", ""), r.get("VGPR_Count", "")))
PY
fi
