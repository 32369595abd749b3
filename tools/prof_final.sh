#!/bin/bash
# kernel-trace statistics of the two training configurations on the final build.  usage: tools/prof_final.sh <tag>
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_train_$tag -- python3 $R/examples/train_maddpg.py --alg maddpg --envs 4096 --episodes 12 > $R/gpurun_out/prof_train_$tag.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_safe_$tag -- python3 $R/examples/train_maddpg.py --alg safemaddpg --envs 8192 --episodes 12 > $R/gpurun_out/prof_safe_$tag.log 2>&1 || exit 1
cd $R
for k in train safe; do
  f=$(find gpurun_out/prof_${k}_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/${tag}_${k}_kernel_stats.csv
  rm -rf gpurun_out/prof_${k}_$tag
done
head -14 gpurun_out/${tag}_train_kernel_stats.csv | cut -d, -f1-5 | cut -c1-150
head -12 gpurun_out/${tag}_safe_kernel_stats.csv | cut -d, -f1-5 | cut -c1-150
