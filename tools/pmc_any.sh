#!/bin/bash
# PMC pass over an arbitrary python tool (counters in their own run, no trace domains).
# usage: tools/pmc_any.sh <tag> "<counters>" "<kernel name filter, |-separated>" <script.py> [args]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
ctrs=$1; shift
filt=$1; shift
script=$1; shift
cd /tmp
rocprofv3 --pmc $ctrs --output-format csv -d $R/gpurun_out/pmca_$tag -- python3 $R/$script "$@" > $R/gpurun_out/pmca_$tag.log 2>&1
cd $R
python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmca_$tag/**/*counter_collection.csv",recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for r in csv.DictReader(open(fn)):
        k=r["Kernel_Name"][:56]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()):
    if any(t in k for t in "$filt".split("|")):
        print(k, {c:(round(sum(x)/len(x),1), len(x)) for c,x in v.items()})
PY
