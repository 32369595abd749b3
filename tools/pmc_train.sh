#!/bin/bash
# PMC pass over the TRAINING loop (the env step kernel with the replay sink, the learner kernels): counters in their own
# run, no trace domains (gpurun refuses the combination).  usage: tools/pmc_train.sh <tag> "<counters>" [train_maddpg args]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
ctrs=$1; shift
cd /tmp
rocprofv3 --pmc $ctrs --output-format csv -d $R/gpurun_out/pmct_$tag -- python3 $R/examples/train_maddpg.py --alg maddpg --envs 4096 --episodes 1 "$@" > $R/gpurun_out/pmct_$tag.log 2>&1
cd $R
python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmct_$tag/**/*counter_collection.csv",recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for r in csv.DictReader(open(fn)):
        k=r["Kernel_Name"][:64]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()):
    if any(t in k for t in ("flex_step", "flex_rollout_burst", "critic_tail", "actor_forward_mfma", "actor_rollout16", "wgrad_kernel", "Cijk")):
        print(k, {c:(round(sum(x)/len(x),1), len(x)) for c,x in v.items()})
PY
