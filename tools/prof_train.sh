#!/bin/bash
# rocprofv3 kernel-trace summaries of the training loop and of the sub-updates (usage: tools/prof_train.sh <tag>)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_train_$tag -- python3 $R/examples/train_maddpg.py --alg maddpg --envs 4096 --episodes 8 > $R/gpurun_out/prof_train_$tag.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_upd_$tag -- python3 $R/tools/update_prof_plain.py > $R/gpurun_out/prof_upd_$tag.log 2>&1
cd $R
for k in train upd; do
  f=$(find gpurun_out/prof_${k}_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/${tag}_${k}_kernel_stats.csv
done
tail -2 gpurun_out/prof_train_$tag.log
tail -4 gpurun_out/prof_upd_$tag.log
