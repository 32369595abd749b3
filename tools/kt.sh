#!/bin/bash
# kernel-trace statistics of a python tool: tools/kt.sh <tag> "<kernel filter>" <script.py> [args]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
filt=$1; shift
script=$1; shift
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_$tag -- python3 $R/$script "$@" > $R/gpurun_out/kt_$tag.log 2>&1 || exit 1
cd $R
f=$(find gpurun_out/kt_$tag -name "*kernel_stats.csv" | head -1)
grep -E "Name|$filt" "$f" | cut -d, -f1-6 | cut -c1-200
