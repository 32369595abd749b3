#!/bin/bash
# kernel statistics of whole update events at the reference's sample reuse (batch = 32 x n_envs): tools/update_prof_plain.py
# with EVENTS / BATCH_DIV under rocprofv3 --kernel-trace --stats.  usage: tools/prof_event.sh <tag> [batch_div]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
div=${2:-1}
cd /tmp
EVENTS=6 BATCH_DIV=$div FILL_EPISODES=2 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_event_$tag -- python3 $R/tools/update_prof_plain.py > $R/gpurun_out/prof_event_$tag.log 2>&1 || { tail -5 $R/gpurun_out/prof_event_$tag.log; exit 1; }
cd $R
f=$(find gpurun_out/prof_event_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${tag}_event_kernel_stats.csv
head -30 gpurun_out/${tag}_event_kernel_stats.csv | cut -c1-150
tail -3 gpurun_out/prof_event_$tag.log
