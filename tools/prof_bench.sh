#!/bin/bash
# rocprofv3 kernel-trace summary of bench.py's env-only leg (usage: tools/prof_bench.sh <tag>)
# (2048 + 256 steps: in the headline form every flex_step_many_kernel launch of the run is 256 steps long)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench_$tag -- python3 $R/bench.py --steps 2048 --warmup 256 --no-cpu-baseline --no-train --no-sustained --no-kernel-shares > $R/gpurun_out/prof_bench_$tag.log 2>&1
cd $R
f=$(find gpurun_out/prof_bench_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${tag}_bench_kernel_stats.csv
tail -1 gpurun_out/prof_bench_$tag.log | cut -c1-300
