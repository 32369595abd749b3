#!/bin/bash
# How the env kernels' rate moves with the number of environments per GPU, on the shipped build (register budget for two
# wavefronts per SIMD: 235 VGPRs) and on a variant of flex_step_many_kernel budgeted for three (168 VGPRs, 85 spilled, 208 B of
# scratch per lane: __launch_bounds__(256, 3)), built here as tools/variants/libflexenv_hip_w3.so (not shipped, not in history).
# Recipe for the variant (in the build container; removed again after the run): change flex_step_many_kernel's
# __launch_bounds__ second argument to 3, hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Iinclude -Isafe-marl_amd/csrc -c
# safe-marl_amd/csrc/flexenv.hip, link with the other objects of safe-marl_amd/build/ into tools/variants/libflexenv_hip_w3.so.
# 4096 envs at two per wavefront are 2048 wavefronts = exactly two per SIMD of 256 CUs.   usage: tools/occupancy_probe.sh <tag>
R=$GRAFT_REPO_ROOT; tag=$1; O=$R/gpurun_out/${tag}_occupancy_probe.txt
cd $R; : > $O
for n in 4096 6144 8192 12288 16384; do
  echo "== shipped build, $n envs" >> $O
  timeout -k 10 120 python3 tools/step_many_bench.py --envs $n --steps 760 --reps 5 >> $O 2>&1 || exit 1
  echo "== three-wavefront budget, $n envs" >> $O
  FLEX_LIB_OVERRIDE=$R/tools/variants/libflexenv_hip_w3.so timeout -k 10 120 python3 tools/step_many_bench.py --envs $n --steps 760 --reps 5 >> $O 2>&1 || exit 1
done
grep -c "step_many" $O
