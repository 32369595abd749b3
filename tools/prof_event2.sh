#!/bin/bash
# kernel trace of whole update events (tools/update_prof_plain.py EVENTS=4) + per-event launch sequence.  usage: tools/prof_event2.sh <tag> [batch_div]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; div=${2:-1}
cd /tmp
EVENTS=4 BATCH_DIV=$div FILL_EPISODES=2 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_event_$tag -- python3 $R/tools/update_prof_plain.py > $R/gpurun_out/prof_event_$tag.log 2>&1 || { tail -5 $R/gpurun_out/prof_event_$tag.log; exit 1; }
cd $R
f=$(find gpurun_out/prof_event_$tag -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 tools/event_sequence.py "$f" --full > gpurun_out/${tag}_event_sequence.txt 2>&1
f2=$(find gpurun_out/prof_event_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f2" ] && cp "$f2" gpurun_out/${tag}_event_kernel_stats.csv
f3=$(find gpurun_out/prof_event_$tag -name "*kernel_trace.csv" | head -1)
[ -n "$f3" ] && python3 tools/update_timeline.py "$f3" > gpurun_out/${tag}_update_timeline.txt 2>&1
grep -v "^ *[0-9.]* us  gap" gpurun_out/${tag}_event_sequence.txt | head -50
tail -3 gpurun_out/prof_event_$tag.log
rm -rf gpurun_out/prof_event_$tag
