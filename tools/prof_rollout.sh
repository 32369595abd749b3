#!/bin/bash
# kernel traces of the training loop (MADDPG 4096 envs, SAFEMADDPG 8192 envs) and the rollout step's kernel sequence inside
# bursts (usage on the GPU box: tools/prof_rollout.sh <tag>)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
O=$R/gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prt_$tag -- python3 $R/examples/train_maddpg.py --alg maddpg --envs 4096 --episodes 12 > $O/prt_$tag.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prs_$tag -- python3 $R/examples/train_maddpg.py --alg safemaddpg --envs 8192 --episodes 12 > $O/prs_$tag.log 2>&1 || exit 1
cd $R
f=$(find gpurun_out/prt_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${tag}_train_kernel_stats.csv
f=$(find gpurun_out/prs_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${tag}_train_safemaddpg_kernel_stats.csv
f=$(find gpurun_out/prt_$tag -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python3 tools/rollout_timeline.py "$f" > $O/${tag}_rollout_timeline.txt 2>&1
f=$(find gpurun_out/prs_$tag -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python3 tools/rollout_seq.py "$f" > $O/${tag}_rollout_seq_safemaddpg.txt 2>&1
head -3 $O/${tag}_rollout_timeline.txt; tail -3 $O/${tag}_rollout_seq_safemaddpg.txt
tail -1 $O/prt_$tag.log | cut -c1-300; tail -1 $O/prs_$tag.log | cut -c1-300
