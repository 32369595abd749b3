#!/bin/bash
# Short trainings over a matrix of configurations (algorithm x envs x agents): every run must finish with finite statistics.
# usage (GPU box): bash tools/config_matrix.sh
cd ${GRAFT_REPO_ROOT:-.}
fail=0
for cfg in "maddpg 64 5" "maddpg 1000 5" "maddpg 2048 3" "maddpg 4096 3" "maddpg 6000 5" "safemaddpg 1000 5" "safemaddpg 4096 5" \
           "matd3 1000 5" "matd3 4096 3" "iddpg 1000 5" "iddpg 4096 3"; do
  set -- $cfg
  out=$(timeout -k 10 200 python examples/train_maddpg.py --alg $1 --envs $2 --agents $3 --episodes 4 2>&1 | tail -1)
  v=$(echo "$out" | python -c "import sys,json,math; d=json.loads(sys.stdin.read()); ok=all(math.isfinite(float(x)) for x in d['stat'].values()); print(('ok' if ok else 'NONFINITE'), round(d['value']/1e6,2), 'M', 'vloss', round(d['stat'].get('mean_train_value_loss', float('nan')),4))" 2>&1 | tail -1)
  echo "$cfg: $v"
  case "$v" in ok*) ;; *) fail=1; echo "$out" | tail -c 600;; esac
done
exit $fail
