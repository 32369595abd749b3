#!/bin/bash
# Round-3 evidence for the env step kernel, both instantiations (SINK = false: bench.py's env-only leg; SINK = true: the
# training loop's rollout): phase stamps from a diagnostic build, PMC passes (one counter set per pass, no trace domains),
# kernel-trace statistics.  usage (on the GPU box): tools/env_profile.sh <tag>
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
O=$R/gpurun_out
cd $R
# 1) phase stamps (diagnostic build of the WHOLE library with -DFLEX_STAMPS: only flexenv.hip looks at the macro)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -DFLEX_STAMPS -Iinclude -Isafe-marl_amd/csrc \
    -o tools/libflexenv_hip_stamps.so safe-marl_amd/csrc/*.hip > $O/${tag}_stamps_build.log 2>&1 || exit 1
FLEX_LIB_OVERRIDE=1 python3 tools/stamps.py 2 > $O/${tag}_stamps_nosink.txt 2>&1 || exit 1
FLEX_LIB_OVERRIDE=1 python3 tools/stamps.py 2 sink > $O/${tag}_stamps_sink.txt 2>&1 || exit 1
echo "stamps done"
# 2) PMC, SINK = false (bench.py env-only leg, eager launches)
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"; do
  name=$(echo $set | cut -d' ' -f1)
  bash tools/pmc.sh ${tag}_nosink_$name "$set" --no-graph > $O/${tag}_pmc_nosink_$name.txt 2>&1 || exit 1
done
echo "pmc nosink done"
# 3) PMC, SINK = true (one training episode after the warm-up episode; graph replays)
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"; do
  name=$(echo $set | cut -d' ' -f1)
  bash tools/pmc_train.sh ${tag}_sink_$name "$set" > $O/${tag}_pmc_sink_$name.txt 2>&1 || exit 1
done
echo "pmc sink done"
# 4) kernel-trace statistics: env-only bench, MADDPG training, SAFEMADDPG training at 8192 envs (BASELINE config 4)
bash tools/prof_bench.sh $tag > $O/${tag}_prof_bench.txt 2>&1 || exit 1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train_$tag -- python3 $R/examples/train_maddpg.py --alg maddpg --envs 4096 --episodes 12 > $O/prof_train_$tag.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_safe_$tag -- python3 $R/examples/train_maddpg.py --alg safemaddpg --envs 8192 --episodes 12 > $O/prof_safe_$tag.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_upd_$tag -- python3 $R/tools/update_prof_plain.py > $O/prof_upd_$tag.log 2>&1 || exit 1
cd $R
for k in train safe upd; do
  f=$(find gpurun_out/prof_${k}_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/${tag}_${k}_kernel_stats.csv
done
# kernel sequences: one fused rollout step inside a burst, one value and one policy sub-update
f=$(find gpurun_out/prof_train_$tag -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 tools/rollout_timeline.py "$f" > $O/${tag}_rollout_timeline.txt 2>&1
f=$(find gpurun_out/prof_safe_$tag -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 tools/rollout_seq.py "$f" > $O/${tag}_rollout_seq_safemaddpg.txt 2>&1
f=$(find gpurun_out/prof_upd_$tag -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 tools/update_timeline.py "$f" > $O/${tag}_update_timeline.txt 2>&1
tail -1 $O/prof_train_$tag.log | cut -c1-400
tail -1 $O/prof_safe_$tag.log | cut -c1-400
echo "traces done"
