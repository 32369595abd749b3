#!/bin/bash
# The evidence set of round 5's second half on the final build: counter passes of the env kernel (digest-stamped), kernel-trace
# statistics of the env-only bench, same-box A/B of an update event (riders + event graph on / off) at both batch sizes, the
# value sub-update's timeline, a 300-episode soak.  usage (on the GPU box): tools/prof_round5b.sh <tag>
R=$GRAFT_REPO_ROOT; tag=$1; O=$R/gpurun_out
cd $R
bash tools/env_counters.sh $tag 4096 > $O/${tag}_counters.log 2>&1 || { tail -5 $O/${tag}_counters.log; exit 1; }
echo "counters done"
bash tools/prof_bench.sh $tag > $O/${tag}_prof_bench.txt 2>&1 || exit 1
echo "prof_bench done"
OFF="FLEX_EVENT_GRAPH=0 FLEX_TD_STATS_RIDER=0 FLEX_WGRAD_FINISH_RIDER=0"
for div in 4 1; do
  bash tools/prof_event2.sh ${tag}_new$div $div > $O/${tag}_new${div}_summary.txt 2>&1 || exit 1
  env $OFF bash tools/prof_event2.sh ${tag}_old$div $div > $O/${tag}_old${div}_summary.txt 2>&1 || exit 1
  bash tools/prof_event2.sh ${tag}_newb$div $div > $O/${tag}_newb${div}_summary.txt 2>&1 || exit 1
done
echo "events done"
timeout -k 10 400 python3 tools/soak.py maddpg 300 > $O/${tag}_soak.txt 2>&1 || { tail -5 $O/${tag}_soak.txt; exit 1; }
tail -2 $O/${tag}_soak.txt
