"""Per-phase cycle breakdown of flex_step_kernel from a DIAGNOSTIC build (-DFLEX_STAMPS, s_memtime stamps).
Build it first:  hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -DFLEX_STAMPS -Iinclude -Isafe-marl_amd/csrc \
                   -o tools/libflexenv_hip_stamps.so safe-marl_amd/csrc/flexenv.hip
Then:            python tools/stamps.py 2      (2 = sweep solver, 0 = Newton+tree).  Never quote its run time."""
import sys, os; sys.path.insert(0,'.')
import numpy as np, torch, ctypes as C
import safe_marl_amd
from safe_marl_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ.get('FLEX_STAMPS_LIB', 'tools/libflexenv_hip_stamps.so'))
from safe_marl_amd.network import create_network
from safe_marl_amd.series import make_synthetic_series
from safe_marl_amd.flex_env import VecFlexProvisionEnv
solver = int(sys.argv[1]) if len(sys.argv)>1 else 2
sink = len(sys.argv) > 2 and sys.argv[2] == "sink"      # the SINK instantiation: the step files its own transition
net=create_network(); s=make_synthetic_series(net,n_days=200)
N=int(os.environ.get("FLEX_STAMPS_N", "4096"))
env=VecFlexProvisionEnv({}, N, net=net, series=s, seed=1, warm_start=True, solver=solver,
                        sweep_accel=os.environ.get('FLEX_NO_SWEEP_ACCEL') != '1')
lib=_lib.load()
stamps=torch.zeros(N,16,dtype=torch.int64,device='cuda')
pool=(0.5+0.5*torch.rand(8,N,5,4,device='cuda')).float()
env.reset()
kw = {}
if sink:
    from safe_marl_amd.replay_buffer import TransReplayBuffer
    buf = TransReplayBuffer(N * 16, device='cuda'); buf.alloc_slabs(N, 5, 144, 4, 64, history=24)
    act_buf = torch.rand(N * 5, 4, device='cuda'); hid_buf = torch.randn(N * 5, 64, device='cuda')
    acc = torch.zeros(N, 10, dtype=torch.float64, device='cuda')
    env.set_obs_ring(buf.cursor[1:], N * 5 * buf.ROW_W, buf.slabs)
    env.set_replay_sink(act_buf, hid_buf, buf.small_ring, buf.hid_ring, acc, cursor_out=buf.cursor[0:1])
    kw = dict(obs_ring=buf.row_ring, replay_sink=True)
    def adv():                      # the policy kernel's part of the cursor protocol: cell 1 = the slab it just read
        buf.cursor[1] = buf.cursor[0]
else:
    def adv(): pass
kw.setdefault('obs_rows', True)
for k in range(30): adv(); env.step(pool[k%8], auto_reset=True, **kw)
lib.flexenv_debug_set_stamps.argtypes=[C.c_void_p]
lib.flexenv_debug_set_stamps(C.c_void_p(stamps.data_ptr()))
adv(); env.step(pool[0], auto_reset=True, **kw)
torch.cuda.synchronize()
st=stamps.cpu().numpy().astype(np.float64)
t0=st[:,0].min()
names=['start->loaded','solve','reward+stores','obs']
d=np.diff(st[:,:5],axis=1)
print('solver',solver,'SINK' if sink else 'no sink','phase cycles (memtime ticks @100MHz?) mean/median/max:')
for i,nm in enumerate(names): print(f'  {nm:16s} {d[:,i].mean():9.1f} {np.median(d[:,i]):9.1f} {d[:,i].max():9.1f}')
rt0, rt1 = st[:,5], st[:,6]
print('  wave lifetime cycles', (st[:,4]-st[:,0]).mean(), ' realtime ticks(100MHz) per wave', (rt1-rt0).mean(), ' => clock GHz', ((st[:,4]-st[:,0])/(rt1-rt0)).mean()*0.1)
print('  kernel span us (realtime)', (rt1.max()-rt0.min())/100.0, ' start spread us', (rt0.max()-rt0.min())/100.0, ' end spread us', (rt1.max()-rt1.min())/100.0)
print('  sweeps', env.peek('PF_SWEEPS').float().mean().item(), 'newton', env.peek('PF_ITERS').float().mean().item())
sw=env.peek('PF_SWEEPS').cpu().numpy(); print('  sweeps hist', np.bincount(sw, minlength=13), ' mean of per-wavefront max', sw.reshape(-1,2).max(1).mean(), ' solve cycles by wavefront max sweeps:', {int(k): round(float(d[::2,1][sw.reshape(-1,2).max(1)==k].mean())) for k in np.unique(sw.reshape(-1,2).max(1))})
rel=(rt0-rt0.min())/100.0
import numpy as np
h,edges=np.histogram(rel,bins=16)
print('start-time histogram (us):'); print(np.round(edges,1)); print(h)
blk=np.arange(N)//4
late=rel>5
print('late blocks: count', late.sum(), 'first late env ids', np.where(late)[0][:16], ' late block ids mod 8 hist', np.bincount((blk[late])%8, minlength=8))
end=(rt1-rt0.min())/100.0
h,edges=np.histogram(end,bins=14)
print('end-time histogram (us):'); print(np.round(edges,1)); print(h)
sw=env.peek('PF_SWEEPS').cpu().numpy(); nt=env.peek('PF_ITERS').cpu().numpy()
print('sweeps hist', np.bincount(sw), 'newton hist', np.bincount(nt))
for lo,hi in ((0,16),(16,20),(20,24),(24,40)):
    m=(end>=lo)&(end<hi)
    if m.sum(): print(f'end in [{lo},{hi}) n={m.sum()} mean sweeps {sw[m].mean():.2f} newton {nt[m].mean():.3f} load {d[m,0].mean():.0f} solve {d[m,1].mean():.0f} cu-block-id mod 256 spread', len(np.unique((np.where(m)[0]//4)%256)))
