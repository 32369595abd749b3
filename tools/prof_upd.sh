#!/bin/bash
# kernel trace of 20 value + 20 policy sub-updates (tools/update_prof_plain.py) and their kernel sequences.  usage: tools/prof_upd.sh <tag>
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_upd_$tag -- python3 $R/tools/update_prof_plain.py > $R/gpurun_out/prof_upd_$tag.log 2>&1 || exit 1
cd $R
f=$(find gpurun_out/prof_upd_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${tag}_upd_kernel_stats.csv
f=$(find gpurun_out/prof_upd_$tag -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 tools/update_timeline.py "$f" > gpurun_out/${tag}_update_timeline.txt 2>&1
tail -45 gpurun_out/${tag}_update_timeline.txt | cut -c1-120
