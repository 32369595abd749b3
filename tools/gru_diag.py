import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from safe_marl_amd import nets
from safe_marl_amd.nets import RNNAgent
from safe_marl_amd.util import convert
G = "tests/golden"
for n, rows in ((8, 4096), (6, 6144), (7, 7168), (4, 4096), (2, 4096)):
    d = json.load(open(os.path.join(G, "learner_args.json"))); d.update(agent_num=n)
    th.manual_seed(5)
    ag = RNNAgent(144 + n, convert(d)).cuda()
    with th.no_grad():
        for p in ag.parameters(): p.mul_(3.0).add_(0.05 * th.randn_like(p))
    g = th.Generator(device="cuda").manual_seed(rows + n)
    obs = 0.5 * th.randn(rows, 144, device="cuda", generator=g); hid = 0.5 * th.randn(rows, 64, device="cuda", generator=g)
    proj = th.randn(rows, 4, device="cuda", generator=g) / rows
    def run(mode):
        ag.fused_training = mode != "autograd"
        nets.GRU_BWD_FUSED = mode == "fused"
        ag.zero_grad()
        means, _, _ = ag.forward_update(obs, hid, n, True)
        nets._DEBUG_KEEP = {}
        (means * proj).sum().backward()
        out = {k: p.grad.double().clone() for k, p in ag.named_parameters()}
        out["_dz"] = nets._DEBUG_KEEP.get("dz")
        nets._DEBUG_KEEP = None
        return out
    ga, gf, gc = run("autograd"), run("fused"), run("composition")
    print(f"n={n} rows={rows}")
    dzf, dzc = gf.pop("_dz"), gc.pop("_dz"); ga.pop("_dz")
    e = (dzf - dzc).abs().amax(1); sc = dzc.abs().max().item()
    bad = (e > 1e-4 * sc).nonzero().flatten()
    print(f"  dz: max err {e.max().item()/sc:.2e}, bad rows {bad.numel()} of {rows}; first bad rows {bad[:24].tolist()}; agents {(bad[:24] % n).tolist()}")
    if bad.numel():
        r = bad[0].item(); print("   row", r, "units with error:", ((dzf[r]-dzc[r]).abs() > 1e-4*sc).nonzero().flatten().tolist()[:32])
    for k in ga:
        s = ga[k].abs().max().item()
        print(f"  {k:16s} fused-vs-autograd {(gf[k]-ga[k]).abs().max().item()/s:.2e}  composition-vs-autograd {(gc[k]-ga[k]).abs().max().item()/s:.2e}  fused-vs-composition {(gf[k]-gc[k]).abs().max().item()/s:.2e}")
