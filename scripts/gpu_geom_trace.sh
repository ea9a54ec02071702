#!/bin/bash
# usage: scripts/gpu_geom_trace.sh <tag> [points] : kernel trace of 4 geometry builds, serial (overlap 0) and overlapped
tag=$1; n=${2:-10000000}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for ov in 0 1; do
  out=/tmp/geomtrace_${tag}_$ov; rm -rf $out
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python $root/scripts/prof_geom.py $n $ov > $root/gpurun_out/${tag}_geom_ov$ov.log 2>&1
  t=$(find $out -name "*kernel_trace.csv" | head -1)
  python3 $root/scripts/geom_timeline.py $t ${MARKER:-k_point_codes} > $root/gpurun_out/${tag}_geom_timeline_ov$ov.txt
  python3 $root/scripts/trace_table.py $t ${MARKER:-k_point_codes} 60 > $root/gpurun_out/${tag}_geom_table_ov$ov.txt
  grep build $root/gpurun_out/${tag}_geom_ov$ov.log
done
