"""us per launch of the replay-window refresh at the update batch (flexnet_gather_window + flexnet_gather_rows): graph-free loop."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import safe_marl_amd
from safe_marl_amd.replay_buffer import TransReplayBuffer
N, bs = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 32768
buf = TransReplayBuffer(N * 192, device="cuda")
buf.alloc_slabs(N, 5, 144, 4, 64, history=24)
buf.row_ring.normal_()
buf.row_ring.view(buf.slabs, N, 5, buf.ROW_W)[..., 6] = 23.0
buf.k, buf.first = 190, 23
win = torch.zeros(bs + N, 720, device="cuda")
hidw = torch.zeros(bs, 320, device="cuda")
def timed(fn, n=200):
    for _ in range(20): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
t_obs = timed(lambda: buf.stacked_obs(40 * N + 17, bs + N, out=win))
t_hid = timed(lambda: buf.gather([("hid_ring", 0, None, N, bs, hidw)], 40 * N + 17))
# check against a plain torch composition
want = torch.zeros_like(win)
rr = buf.row_ring.view(buf.slabs, N, 5, 8)
slot0 = 40 * N + 17
rows = torch.arange(bs + N, device="cuda") + slot0
sl, env = (rows // N) % buf.slabs, rows % N
for h in range(24):
    src = (sl - (23 - h)) % buf.slabs
    want.view(bs + N, 5, 24, 6)[:, :, h] = rr[src, env][:, :, :6]
print(f"gather_window {t_obs:.1f} us ({(bs + N) * 2880 / t_obs / 1e6:.2f} TB/s written), hid rows {t_hid:.1f} us, equal: {torch.equal(win, want)}")
