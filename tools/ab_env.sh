#!/bin/bash
# A/B legs of the env-only bench (usage on the GPU box: tools/ab_env.sh <tag>): sweep acceleration on/off, sweep_tol fraction
R=$GRAFT_REPO_ROOT
tag=$1
B="python bench.py --no-train --no-cpu-baseline --steps 512 --warmup 64"
pick() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c=d["config"]; s=d.get("sustained") or {}
print("%-28s value %.1f M  sustained %.1f M  dev %.3f us  sweeps %.3f  newton %.4f  failed %.4f" % (sys.argv[1].split("/")[-1], d["value"]/1e6, (s.get("value") or 0)/1e6, 1e3*d["roofline"]["avg_launch_ms"], c["pf_sweeps_mean"], c["pf_newton_iters_mean"], c["solver_failed_frac"]))
PY
}
$B > gpurun_out/${tag}_ab_accel.json 2>gpurun_out/${tag}_ab_accel.err && pick gpurun_out/${tag}_ab_accel.json &&
$B --no-sweep-accel > gpurun_out/${tag}_ab_plain.json 2>gpurun_out/${tag}_ab_plain.err && pick gpurun_out/${tag}_ab_plain.json &&
FLEX_SWEEP_TOL_FRAC=0.25 $B > gpurun_out/${tag}_ab_accel_frac25.json 2>gpurun_out/${tag}_ab_frac25.err && pick gpurun_out/${tag}_ab_accel_frac25.json &&
FLEX_SWEEP_TOL_FRAC=0.25 $B --no-sweep-accel > gpurun_out/${tag}_ab_plain_frac25.json 2>gpurun_out/${tag}_ab_plain_frac25.err && pick gpurun_out/${tag}_ab_plain_frac25.json &&
$B > gpurun_out/${tag}_ab_accel2.json 2>gpurun_out/${tag}_ab_accel2.err && pick gpurun_out/${tag}_ab_accel2.json
