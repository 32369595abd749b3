#!/bin/bash
# PMC pass over bench.py's env-only leg (counters in their own run, no trace domains: gpurun refuses the combination).
# usage: tools/pmc.sh <tag> "<counters>" [bench args]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
ctrs=$1; shift
cd /tmp
rocprofv3 --pmc $ctrs --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-train --no-sustained --no-kernel-shares "$@" > $R/gpurun_out/pmc_$tag.log 2>&1
cd $R
python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmc_$tag/**/*counter_collection.csv",recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for r in csv.DictReader(open(fn)):
        k=r["Kernel_Name"][:40]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    if "flex" in k:
        print(k, {c:(sum(x)/len(x), len(x)) for c,x in v.items()})
PY
