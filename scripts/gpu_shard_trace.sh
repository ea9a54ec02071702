#!/bin/bash
# usage: scripts/gpu_shard_trace.sh <tag> <world> <rank> <shard_geometry> [points]: kernel trace of one rank's sharded forward
# with a transport that moves nothing (scripts/shard_dry_run.py); timeline of the last forward with the idle gaps
tag=$1; world=$2; rank=$3; geom=$4; n=${5:-10000000}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=/tmp/shardtrace_$tag; rm -rf $out
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python $root/scripts/shard_dry_run.py $n $world $rank $geom > $root/gpurun_out/${tag}_shard.log 2>&1
t=$(find $out -name "*kernel_trace.csv" | head -1)
python3 $root/scripts/geom_timeline.py $t ${MARKER:-k_point_codes} > $root/gpurun_out/${tag}_shard_timeline.txt
tail -3 $root/gpurun_out/${tag}_shard.log
tail -1 $root/gpurun_out/${tag}_shard_timeline.txt
