"""How the OPF comparator's outer (sequential convex programming) iteration converges, per power-flow solver behind its
central-difference sensitivities: max control move per outer iteration.  python tools/opf_outer_probe.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import safe_marl_amd  # noqa: F401
from safe_marl_amd import _lib, opf as opf_mod, flex_env
from safe_marl_amd.network import create_network
from safe_marl_amd.series import make_synthetic_series

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
net = create_network()
tab = np.asarray(make_synthetic_series(net, n_days=400).table)
rows = np.stack([tab[96 * (3 + b):96 * (3 + b) + 96] for b in range(B)])
orig = flex_env.pf_solve_batch
for name, kw in (("sweep (default)", {}), ("tree Newton", dict(solver=_lib.FLEX_SOLVER_TREE))):
    opf_mod.pf_solve_batch = lambda net_, p, q, want_branch=False, _kw=kw: orig(net_, p, q, want_branch=want_branch, **_kw)
    r = opf_mod.BatchedOPF(net).solve(rows[:, :, 71], rows[:, :, :33], rows[:, :, 33:66], rows[:, :, 66:71], np.full((B, 5), 0.0125))
    torch.cuda.synchronize()
    print(f"{name:18s} outer {r['outer_iters']:2d}  objective {r['objective'].mean().item():+.9f}  moves " +
          " ".join(f"{h['move']:.1e}" for h in r["history"]) + "  ipm " + " ".join(str(int(h["ipm_iters"])) for h in r["history"]))
