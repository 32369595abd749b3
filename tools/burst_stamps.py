"""Phase breakdown of flex_rollout_burst_kernel from a DIAGNOSTIC build (-DFLEX_STAMPS):
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -DFLEX_STAMPS -Iinclude -Isafe-marl_amd/csrc \
        -o tools/libflexenv_burst_stamps.so safe-marl_amd/csrc/*.hip
  python tools/burst_stamps.py [n_envs]
Per wavefront, last step of a 16-step burst: policy phase, wait at the first hand-over, environment phase, wait at the second.
Never quote its run time."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from safe_marl_amd import _lib

_lib.LIB_PATH = os.path.abspath("tools/libflexenv_burst_stamps.so")


def main():
    n_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    from test_rollout_gpu import _trainer
    from safe_marl_amd.learner import RolloutGraph
    tr = _trainer(n_envs)
    rg = RolloutGraph(tr.behaviour_net, tr.env, tr.replay_buffer)
    rg.start_episode(tr.env.reset())
    rg.capture()
    rg.run(40)
    lib = _lib.load()
    stamps = torch.zeros(n_envs, 16, dtype=torch.int64, device="cuda")
    lib.flexenv_debug_set_stamps.argtypes = [C.c_void_p]
    lib.flexenv_debug_set_stamps(C.c_void_p(stamps.data_ptr()))
    rg.body(burst=16)                                         # eager: the stamps pointer is read when the launch is made
    torch.cuda.synchronize()
    rg.buf.k += 16
    st = stamps.cpu().numpy().astype(np.float64)[0::2]        # one row per wavefront
    wave = np.arange(st.shape[0]) % 8
    ph = {"policy (s12 -> s8)": st[:, 8] - st[:, 12], "wait 1": st[:, 9] - st[:, 8], "env": st[:, 10] - st[:, 9],
          "wait 2": st[:, 11] - st[:, 10], "step": st[:, 11] - st[:, 12]}
    env_ph = {"env: start->loaded": st[:, 1] - st[:, 0], "env: solve": st[:, 2] - st[:, 1], "env: reward+stores": st[:, 3] - st[:, 2],
              "env: obs": st[:, 4] - st[:, 3], "env: handover -> start": st[:, 0] - st[:, 9], "env: end -> s10": st[:, 10] - st[:, 4]}
    print("ticks of s_memtime (100 MHz): mean / median / max, then the mean per wavefront slot 0-7")
    for name, d in list(ph.items()) + list(env_ph.items()):
        per = " ".join(f"{d[wave == w].mean():7.1f}" for w in range(8))
        print(f"  {name:24s} {d.mean():8.1f} {np.median(d):8.1f} {d.max():8.1f}   | {per}")


if __name__ == "__main__":
    main()
