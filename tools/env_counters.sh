#!/bin/bash
# Counter passes over bench.py's env-only leg -> gpurun_out/<tag>_pmc_traffic.json (copy it to profiles/pmc_traffic.json:
# bench.py reports roofline.traffic / valu_issue_frac / limiter from it, and only when its source_digest is the build's).
# One counter set per pass, --pmc only (no trace domains: gpurun refuses the combination); FETCH_SIZE and WRITE_SIZE in
# separate passes (TCC slots), the gfx950 FETCH_SIZE half-count corrected in the post-processor (MI355X_MICROARCH.md §HBM).
# usage (on the GPU box): tools/env_counters.sh <tag> [envs]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
envs=${2:-4096}
O=$R/gpurun_out
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  cd /tmp
  # (headline form: flexenv_step_many — every launch of these passes is 20 steps long: 1 set-up + 1 warm-up + 3 timed + 12 bracketed)
  rocprofv3 --pmc $set --output-format csv -d $O/ctr_${tag}_$i -- python3 $R/bench.py --steps 60 --warmup 20 --steps-per-launch ${SPL:-20} --envs $envs \
      ${FORM:+--launch-form $FORM} --no-cpu-baseline --no-train --no-sustained --no-kernel-shares --no-graph > $O/ctr_${tag}_$i.log 2>&1 || { tail -5 $O/ctr_${tag}_$i.log; exit 1; }
  cd $R
done
python3 tools/env_counters_post.py $tag $envs ${SPL:-20} ${FORM:-many}
