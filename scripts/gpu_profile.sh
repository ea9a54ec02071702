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
