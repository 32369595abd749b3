"""Host-side logic around the learner kernels that must behave identically without a GPU: every fused path declines
on CPU tensors and the PyTorch composition it stands in for gives the same numbers (the GPU suites pin the kernels)."""
import os
import sys
import types

import numpy as np
import torch


def _alg_args(**over):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from train_maddpg import DEFAULT_ALG_ARGS
    from safe_marl_amd.util import convert
    a = dict(DEFAULT_ALG_ARGS)
    a.update(alg="maddpg", agent_num=3, obs_size=12, state_size=20, action_dim=4, cuda=False)
    a.update(over)
    return convert(a)


def test_clip_and_step_falls_back_to_pytorch_on_cpu():
    from safe_marl_amd.optim import clip_and_step
    from torch.optim import RMSprop
    torch.manual_seed(0)
    a = [torch.nn.Parameter(torch.randn(7, 5)), torch.nn.Parameter(torch.randn(5))]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    oa, ob = RMSprop(a, lr=1e-2, alpha=0.99, eps=1e-5), RMSprop(b, lr=1e-2, alpha=0.99, eps=1e-5)
    for step in range(3):
        for pa, pb in zip(a, b):
            g = torch.randn_like(pa) * 4
            pa.grad, pb.grad = g.clone(), g.clone()
        na = clip_and_step(oa, a, 1.0)
        nb = torch.nn.utils.clip_grad_norm_(b, 1.0)
        ob.step()
        assert torch.equal(na, nb)
        for pa, pb in zip(a, b):
            assert torch.equal(pa, pb)


def test_fused_predicates_decline_cpu_tensors():
    from safe_marl_amd import nets
    args = types.SimpleNamespace(hid_size=64, layernorm=True, hid_activation="relu", action_dim=4, agent_num=3)
    agent = nets.RNNAgent(12 + 3, args)
    critic = nets.MLPCritic(3 * 12 + 3 + 3 * 4, 1, args)
    x = torch.randn(4096, 64)
    assert not nets.tall_wgrad_supported(x, x)
    assert not nets.critic_tail_supported(critic, x)
    assert not nets.critic_replayed_supported(critic, torch.randn(4096, 36), torch.randn(4096, 12), 3)
    assert not nets.td_loss_supported(torch.randn(8, 3), torch.randn(8, 3), torch.randn(8, 3), torch.randn(8), None)
    assert agent.forward_update(torch.randn(12, 12), torch.zeros(12, 64), 3, True) is None
    assert nets.fused_actor_forward(agent, torch.randn(4, 3, 12), torch.zeros(4, 3, 64), 3, True) is None
    # tall_linear / wide_batch_linear are plain linear layers here, gradients included
    w = torch.randn(64, 64, requires_grad=True)
    xs = torch.randn(4096, 64, requires_grad=True)
    ref = torch.autograd.grad(torch.nn.functional.linear(xs, w).sum(), [xs, w])
    got = torch.autograd.grad(nets.tall_linear(xs, w).sum(), [xs, w])
    assert all(torch.equal(a, b) for a, b in zip(ref, got))
    gw, = torch.autograd.grad(nets.wide_batch_linear(xs.detach(), w).pow(2).sum(), [w])
    rw, = torch.autograd.grad((xs.detach() @ w.t()).pow(2).sum(), [w])
    assert torch.allclose(gw, rw, rtol=1e-4, atol=1e-3)


def test_constant_action_avail_mask_is_the_identity_it_replaces():
    """get_actions skips the restore mask when the replay hands out action_avail as the known constant 1.0."""
    from safe_marl_amd.learner import MADDPG
    torch.manual_seed(1)
    m = MADDPG(_alg_args())
    obs = torch.randn(6, 3, 12)
    hid = torch.zeros(6, 3, 64)
    ones = torch.ones(6, 3, 4)
    const = torch.full((1, 1, 1), 1.0).expand(6, 3, 4)
    const._flex_const = 1.0
    with torch.no_grad():
        a1 = m.get_actions(obs, status="train", exploration=False, actions_avail=ones, last_hid=hid)
        a2 = m.get_actions(obs, status="train", exploration=False, actions_avail=const, last_hid=hid)
    assert torch.equal(a1[1], a2[1]) and torch.equal(a1[0], a2[0])
    means, log_stds, _ = m.policy(obs, last_hid=hid)
    assert log_stds.shape == means.shape and float(log_stds.max()) == float(np.log(m.args.fixed_policy_std))


import pytest


@pytest.mark.parametrize("alg", ["maddpg", "matd3", "iddpg"])
def test_update_fields_cover_what_each_loss_reads(alg):
    """MADDPG.update_fields lists the replay fields a graphed sub-update refreshes: poisoning every OTHER field must not
    change the loss (reward is read by both through the BatchNorm running statistics)."""
    from safe_marl_amd import learner
    from safe_marl_amd.replay_buffer import Transition
    torch.manual_seed(2)
    cls = {"maddpg": learner.MADDPG, "matd3": learner.MATD3, "iddpg": learner.IDDPG}[alg]
    m = cls(_alg_args(alg=alg), cls(_alg_args(alg=alg)))
    b = 16
    g = torch.Generator().manual_seed(3)
    base = dict(state=torch.randn(b, 3, 12, generator=g), action=torch.rand(b, 3, 4, generator=g),
                log_prob_a=torch.zeros(b, 3, 4), value=torch.zeros(b, 3, 1), next_value=torch.zeros(b, 3, 1),
                reward=torch.randn(b, 3, generator=g), next_state=torch.randn(b, 3, 12, generator=g),
                done=torch.zeros(b), last_step=torch.zeros(b), action_avail=torch.ones(b, 3, 4),
                last_hid=torch.randn(b, 3, 64, generator=g), hid=torch.randn(b, 3, 64, generator=g))
    stored = ("state", "action", "reward", "next_state", "done", "last_step", "last_hid", "hid")
    for which in ("value", "policy"):
        losses = []
        for poison in (False, True):
            f = {k: v.clone() for k, v in base.items()}
            if poison:
                for k in stored:
                    if k not in m.update_fields[which]:
                        f[k] = torch.full_like(f[k], 123.0)
            torch.manual_seed(9)                        # MATD3 smooths its target with noise
            p, v, _ = m.get_loss(Transition(**f), need=which)
            losses.append((v if which == "value" else p).item())
        assert losses[0] == losses[1], which


def test_stacked_ring_bookkeeping_expands_every_slab_once_and_mirrors_the_head():
    """replay_buffer.enable_stacked_ring / expand_stacked (round 5) on CPU tensors, with flexnet_gather_window replaced by a
    NumPy restatement of include/flexnet.h's definition (dst[i][a][h*6+f] = row_ring[slab - (H-1-h)][env][a][f] while
    H-1-h <= older, else 0): after every batch of vector steps the ring holds each expanded slab's stacked observations at
    its physical position, the first `tail` rows are mirrored behind the ring's end (also for the hidden-state ring), slabs are
    expanded once (the restatement counts), a window that starts anywhere is contiguous, and a longer tail makes a new ring
    generation."""
    from safe_marl_amd.replay_buffer import DeviceReplayBuffer
    N, n, H = 4, 2, 3
    buf = DeviceReplayBuffer(N * 10, device="cpu")
    buf.alloc_slabs(N, n, 6 * H, 4, 8, history=H)                  # 10 + 2 slabs
    S = buf.slabs
    calls = []

    def stacked_obs(slot, rows, out=None):
        calls.append((slot, rows))
        rr = buf.row_ring.view(S, N, n, buf.ROW_W).numpy()
        res = np.zeros((rows, n, H * 6), np.float32)
        for i in range(rows):
            g = slot + i
            slab, env = (g // N) % S, g % N
            for a in range(n):
                older = int(rr[slab, env, a, 6])
                for h in range(H):
                    back = H - 1 - h
                    if back <= older:
                        res[i, a, h * 6:(h + 1) * 6] = rr[(slab - back) % S, env, a, :6]
        t = torch.from_numpy(res.reshape(rows, -1))
        if out is None:
            return t
        out.copy_(t)
        return out

    buf.stacked_obs = stacked_obs
    buf.enable_stacked_ring(3 * N + N)
    assert buf.stack_rows == S * N and buf.stack_tail == 4 * N and buf.stack_gen == 1
    rng = np.random.default_rng(0)
    first = torch.from_numpy(rng.normal(size=(N, n, 6 * H)).astype(np.float32))
    buf.begin_stream(first)
    step = [0]

    def roll(m):
        for _ in range(m):
            p = (buf.k + 1) % S                                   # the step files the NEXT slab's record and hidden state
            rec = buf.row_ring[p].view(N, n, buf.ROW_W)
            rec[:, :, :6] = torch.from_numpy(rng.normal(size=(N, n, 6)).astype(np.float32))
            step[0] += 1
            rec[:, :, 6] = float(min(step[0], H - 1))
            buf.hid_ring[p] = torch.from_numpy(rng.normal(size=(N, n * 8)).astype(np.float32))
            buf.stepped()

    def check():
        lo = buf.first
        for c in range(lo, buf.k + 1):
            p = c % S
            want = stacked_obs(c * N, N)
            assert torch.equal(buf.stack_ring[p * N:(p + 1) * N], want), c
            if p * N < buf.stack_tail:
                assert torch.equal(buf.stack_ring[buf.stack_rows + p * N:buf.stack_rows + (p + 1) * N], want), c
            if p < buf.hid_tail_slabs:
                assert torch.equal(buf.hid_store[S + p], buf.hid_ring[p]), c

    roll(5)
    calls.clear()
    buf.expand_stacked()
    assert sum(r for _, r in calls) == 6 * N                       # slabs 0..5, the cursor's included, once
    check()
    calls.clear()
    buf.expand_stacked()
    assert calls == []                                             # nothing new
    roll(9)                                                        # the ring (12 slabs) wraps
    calls.clear()
    buf.expand_stacked()
    assert sum(r for _, r in calls) == 9 * N
    calls.clear()
    check()
    # a window astride the seam is contiguous in the ring + tail
    slot = (buf.k - 3) * N + 1
    p = slot % buf.stack_rows
    got = buf.stack_ring[p:p + 3 * N]
    assert torch.equal(got, stacked_obs(slot, 3 * N))
    # a longer tail: a new ring and a new generation; everything is expanded again
    buf.enable_stacked_ring(6 * N)
    assert buf.stack_gen == 2 and buf.stack_tail == 6 * N and buf.stacked_next is None
    buf.expand_stacked()
    check()
