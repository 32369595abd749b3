#!/bin/bash
# phase stamps of the env step with and without the sweep extrapolation (usage on the GPU box: tools/stamps_ab.sh <tag>)
R=$GRAFT_REPO_ROOT; tag=$1; O=$R/gpurun_out; cd $R
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -DFLEX_STAMPS -Iinclude -Isafe-marl_amd/csrc \
    -o tools/libflexenv_hip_stamps.so safe-marl_amd/csrc/*.hip > $O/${tag}_stamps_build.log 2>&1 || exit 1
FLEX_LIB_OVERRIDE=1 python3 tools/stamps.py 2 > $O/${tag}_stamps_accel.txt 2>&1 || exit 1
FLEX_LIB_OVERRIDE=1 FLEX_NO_SWEEP_ACCEL=1 python3 tools/stamps.py 2 > $O/${tag}_stamps_plain.txt 2>&1 || exit 1
head -12 $O/${tag}_stamps_accel.txt; head -12 $O/${tag}_stamps_plain.txt
