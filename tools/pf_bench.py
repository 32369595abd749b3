"""Power-flow solver variants through pf_solve_batch (sweep / Newton+tree / Newton+dense LU); run under
rocprofv3 --kernel-trace --stats for kernel-only durations (profiles/r01_pf_solver_variants.txt)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
import safe_marl_amd
from safe_marl_amd.network import create_network
from safe_marl_amd.flex_env import pf_solve_batch
net=create_network()
buses=net['bus_numbers']
p=np.array([net['active_power_demand'][b] for b in buses]); q=np.array([net['reactive_power_demand'][b] for b in buses])
rng=np.random.default_rng(0); n=4096
P=torch.from_numpy(p[None]*rng.uniform(0.5,1.2,(n,33))).cuda(); Q=torch.from_numpy(q[None]*rng.uniform(0.5,1.2,(n,33))).cuda()
for name,solver in (("sweep+verify",2),("newton tree",0),("newton dense LU (LDS)",1)):
    for _ in range(5): out=pf_solve_batch(net,P,Q,solver=solver)
    torch.cuda.synchronize(); t=time.perf_counter(); K=50
    for _ in range(K): out=pf_solve_batch(net,P,Q,solver=solver)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/K
    it=out['iters'].float().mean().item()
    print(f"{name:24s} {dt*1e6:9.1f} us / 4096 flat-start solves  {n/dt/1e6:8.2f} M solves/s  iters(mean, newton + 1000*sweeps) {it:.1f}")
