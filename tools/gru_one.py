"""A few fused actor training passes (forward + backward of nets._ActorTrainFn) at 163 840 rows, for a counter pass or a
kernel trace (tools/pmc_any.sh)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from safe_marl_amd.nets import RNNAgent
from safe_marl_amd.util import convert
n, rows = 5, 163840
d = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/learner_args.json"))); d.update(agent_num=n)
ag = RNNAgent(144 + n, convert(d)).cuda()
obs = 0.5 * th.randn(rows, 144, device="cuda"); hid = 0.5 * th.randn(rows, 64, device="cuda")
proj = th.randn(rows, 4, device="cuda") / rows
for _ in range(6):
    ag.zero_grad()
    means, _, _ = ag.forward_update(obs, hid, n, True)
    (means * proj).sum().backward()
th.cuda.synchronize()
