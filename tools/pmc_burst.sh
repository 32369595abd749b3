#!/bin/bash
# PMC pass over tools/burst_bench.py (burst launch vs policy + environment launches per step): counters in their own run, no
# trace domains.  usage: tools/pmc_burst.sh <tag> "<counters>"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
ctrs=$1; shift
cd /tmp
rocprofv3 --pmc $ctrs --output-format csv -d $R/gpurun_out/pmcb_$tag -- python3 $R/tools/burst_bench.py 4096 5 short > $R/gpurun_out/pmcb_$tag.log 2>&1
cd $R
python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmcb_$tag/**/*counter_collection.csv",recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for r in csv.DictReader(open(fn)):
        k=r["Kernel_Name"][:48]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()):
    if any(t in k for t in ("flex_step", "flex_rollout_burst", "actor_rollout16")):
        print(k, {c:(round(sum(x)/len(x),1), len(x)) for c,x in v.items()})
PY
