"""In-tree build of the HIP library (gfx950 only): hipcc -> safe-marl_amd/libflexenv_hip.so.
The .so is git-ignored but travels to the GPU box with the gpurun snapshot."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = [os.path.join(HERE, "csrc", "flexenv.hip"), os.path.join(HERE, "csrc", "actor.hip"),
       os.path.join(HERE, "csrc", "critic.hip"), os.path.join(HERE, "csrc", "rollout.hip"),
       os.path.join(HERE, "csrc", "wgrad.hip"), os.path.join(HERE, "csrc", "lnrelu.hip"),
       os.path.join(HERE, "csrc", "optim.hip"), os.path.join(HERE, "csrc", "tdloss.hip")]
DEPS = SRC + [os.path.join(HERE, "csrc", "flex_device.h"), os.path.join(HERE, "csrc", "flex_reduce.h"),
              os.path.join(HERE, "csrc", "flex_launch.h"), os.path.join(ROOT, "include", "flexenv.h"),
              os.path.join(ROOT, "include", "flexnet.h")]
OUT = os.path.join(HERE, "libflexenv_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC",
         "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc")]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    cmd = [HIPCC] + FLAGS + ["-o", OUT] + SRC
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
