#!/bin/bash
# rocprofv3 kernel trace of the DRIVER's bench command form (--steps 20 --warmup 5; env-only legs): the flex_step_many_kernel
# launches listed one by one, so that the 20-step launches of the timed region can be read next to bench.py's own
# roofline.avg_launch_ms (the --stats average mixes the set-up launches of other lengths).  usage: tools/prof_bench20.sh <tag>
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench20_$tag -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-kernel-shares > $R/gpurun_out/prof_bench20_$tag.log 2>&1 || { tail -5 $R/gpurun_out/prof_bench20_$tag.log; exit 1; }
cd $R
python3 - $tag <<'PY'
import csv, glob, json, sys
tag = sys.argv[1]
f = glob.glob(f"gpurun_out/prof_bench20_{tag}/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "flex_step_many_kernel" in r["Kernel_Name"]]
d = sorted(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Start_Timestamp"])) for r in rows)
line = json.loads([l for l in open(f"gpurun_out/prof_bench20_{tag}.log").read().splitlines() if l.startswith('{"metric"')][-1])
out = [f"rocprofv3 --kernel-trace of `python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-kernel-shares` ({tag})",
       f"flex_step_many_kernel: {len(d)} launches; durations in us, clustered by launch length:"]
clusters = {}
for us, _ in d:
    key = "5-step (warm-up)" if us < 80 else "20-step (timed region, event-bracketed repeats)" if us < 400 else "256-step (sustained leg, set-up)"
    clusters.setdefault(key, []).append(us)
for k, v in clusters.items():
    out.append(f"  {k:52s} n = {len(v):3d}   mean {sum(v) / len(v):9.2f}   min {min(v):9.2f}   max {max(v):9.2f}")
r = line["roofline"]
out.append(f"bench.py's own line in this run: value {line['value'] / 1e6:.1f} M env-steps/s, roofline.avg_launch_ms {r['avg_launch_ms']:.5f} "
           f"({r['steps_per_launch']} steps per launch), frac {r['frac']:.4f}  (under the profiler)")
open(f"gpurun_out/{tag}_bench20_kernel_trace_summary.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
fs=$(find gpurun_out/prof_bench20_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$fs" ] && cp "$fs" gpurun_out/${tag}_bench20_kernel_stats.csv
rm -rf gpurun_out/prof_bench20_$tag
