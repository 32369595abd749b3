#!/bin/bash
# same-box A/B of env-kernel builds: tools/libflexenv_hip_var{A,B,...}.so (FLEX_LIB_OVERRIDE), env-only bench, three rounds interleaved
# usage on the GPU box: tools/ab_variants.sh <tag> A B C ...
R=$GRAFT_REPO_ROOT; tag=$1; shift
B="python bench.py --no-train --no-cpu-baseline --no-kernel-shares --steps 512 --warmup 64"
for round in 1 2 3; do
  for v in "$@"; do
    FLEX_LIB_OVERRIDE=$R/tools/libflexenv_hip_var$v.so $B > gpurun_out/${tag}_var${v}_$round.json 2>gpurun_out/${tag}_var.err || { tail -3 gpurun_out/${tag}_var.err; exit 1; }
    python - gpurun_out/${tag}_var${v}_$round.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
s=d.get("sustained") or {}
print("var %s  value %.1f M  sustained %.1f M  kernel %.3f us  sweeps %.2f  tol1e-6 %.1f M  newton-sibling %.1f M" % (sys.argv[2], d["value"]/1e6, (s.get("value") or 0)/1e6, 1e3*d["roofline"]["avg_launch_ms"], d["config"]["pf_sweeps_mean"], d["tolerance_sibling"]["value"]/1e6, d["solver_sibling"]["value"]/1e6))
PY
  done
done
