"""The fused actor inference kernel (csrc/actor.hip, include/flexnet.h) against the PyTorch module it replaces,
which carries the reference's parameter names and arithmetic (rnn_agent.py:13-33; golden parity of that module with
the imported reference is tests/test_learner_cpu.py).  fp32; tolerance 2e-5 absolute on O(1) activations (the
summation order differs from rocBLAS)."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


def _agent(obs_dim, n_agents, act_dim, layernorm=True, agent_id=True, seed=0):
    from safe_marl_amd.nets import RNNAgent
    torch.manual_seed(seed)
    args = types.SimpleNamespace(hid_size=64, layernorm=layernorm, action_dim=act_dim, agent_num=n_agents,
                                 hid_activation="relu")
    agent = RNNAgent(obs_dim + (n_agents if agent_id else 0), args).cuda()
    with torch.no_grad():                      # not the tiny init of init_std: make every term matter
        for p in agent.parameters():
            p.copy_(torch.randn_like(p) * 0.3)
    return agent


def _reference(agent, obs, hidden, n_agents, agent_id):
    b = obs.shape[0]
    x = obs
    if agent_id:
        x = torch.cat((obs, torch.eye(n_agents, device=obs.device).expand(b, -1, -1)), -1)     # model.py:105-108
    with torch.no_grad():
        means, _, h = agent(x.reshape(b * n_agents, -1), hidden.reshape(b * n_agents, -1))
    return means, h


@pytest.mark.parametrize("b,n,obs_dim,act,ln,aid", [(4096, 5, 144, 4, True, True), (3, 5, 144, 4, True, True),
                                                    (1, 1, 144, 4, True, False), (37, 3, 72, 2, False, True),
                                                    (130, 8, 6, 8, True, True), (1001, 5, 144, 4, True, True),
                                                    (8192, 5, 144, 4, True, True), (4096, 3, 144, 4, True, True)])
# 0: matrix-core kernels as the library picks them (five 16-row tiles per CU up to 80 rows per CU, 32-row tiles above),
# 1: VALU kernel, 2: the five-tiles-per-CU kernel whatever the size (40 960 rows: two rounds per CU), 3: the 32-row kernel
@pytest.mark.parametrize("variant", [0, 1, 2, 3])
def test_matches_module(b, n, obs_dim, act, ln, aid, variant):
    from safe_marl_amd.nets import fused_actor_forward
    agent = _agent(obs_dim, n, act, layernorm=ln, agent_id=aid)
    g = torch.Generator(device="cuda").manual_seed(1)
    obs = torch.randn(b, n, obs_dim, device="cuda", generator=g)
    hid = torch.randn(b, n, 64, device="cuda", generator=g)
    out = fused_actor_forward(agent, obs, hid, n, aid, variant=variant)
    assert out is not None
    ref_m, ref_h = _reference(agent, obs, hid, n, aid)
    assert torch.isfinite(out[0]).all() and torch.isfinite(out[1]).all()
    assert (out[1] - ref_h).abs().max().item() < 2e-5
    assert (out[0] - ref_m).abs().max().item() < 5e-5 * max(1.0, ref_m.abs().max().item())


def test_unsupported_shapes_fall_back():
    from safe_marl_amd.nets import fused_actor_forward
    agent = _agent(200, 5, 4)                                             # obs wider than the kernel stages
    obs = torch.randn(2, 5, 200, device="cuda")
    assert fused_actor_forward(agent, obs, torch.zeros(2, 5, 64, device="cuda"), 5, True) is None


def test_policy_uses_the_kernel_without_grad_and_the_module_with():
    """Model.policy: same numbers either way; the fused path only where no graph is needed."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from train_maddpg import DEFAULT_ALG_ARGS
    from safe_marl_amd.learner import MADDPG
    from safe_marl_amd.util import convert
    alg = dict(DEFAULT_ALG_ARGS)
    alg.update(alg="maddpg", agent_num=5, obs_size=144, state_size=110, action_dim=4)
    torch.manual_seed(3)
    m = MADDPG(convert(alg)).cuda()
    obs = torch.randn(64, 5, 144, device="cuda")
    hid = torch.randn(64, 5, 64, device="cuda")
    with torch.no_grad():
        f_means, _, f_hid = m.policy(obs, last_hid=hid)
        m.fused_inference = False
        t_means, _, t_hid = m.policy(obs, last_hid=hid)
        m.fused_inference = True
    assert (f_means - t_means).abs().max().item() < 1e-5 and (f_hid - t_hid).abs().max().item() < 1e-5
    g_means, _, _ = m.policy(obs, last_hid=hid)                            # grad enabled: the module, with a graph
    assert g_means.requires_grad


def _agent_and_inputs(rows_b, act_dim):
    agent = _agent(144, 5, act_dim)
    g = torch.Generator(device="cuda").manual_seed(17)
    obs = torch.randn(rows_b, 5, 144, device="cuda", generator=g)
    hid = torch.randn(rows_b, 5, 64, device="cuda", generator=g)
    return agent, obs, hid


def _noise_restatement(seed, step, rows, act_dim):
    """csrc/actor.hip actor_noise4 in NumPy: Philox4x32-10 (oracle.env_oracle), counter (row, group, step lo,
    tag ^ step hi), key = seed; 24-bit uniforms (x + 0.5) / 2^24; Box-Muller."""
    import numpy as np
    from oracle.env_oracle import philox4x32_10
    z = np.zeros((rows, act_dim))
    for row in range(rows):
        for group in range((act_dim + 3) // 4):
            x = philox4x32_10((row, group, step & 0xFFFFFFFF, 0xAC70A5E1 ^ (step >> 32)), (seed & 0xFFFFFFFF, seed >> 32))
            u = [((v >> 8) + 0.5) / 16777216.0 for v in x]
            ra, rb = np.sqrt(-2.0 * np.log(u[0])), np.sqrt(-2.0 * np.log(u[2]))
            four = [ra * np.cos(2 * np.pi * u[1]), ra * np.sin(2 * np.pi * u[1]),
                    rb * np.cos(2 * np.pi * u[3]), rb * np.sin(2 * np.pi * u[3])]
            for r in range(4):
                if 4 * group + r < act_dim:
                    z[row, 4 * group + r] = four[r]
    return z


@pytest.mark.parametrize("act_dim,step", [(4, 0), (4, 123456789012), (6, 7)])
def test_in_kernel_exploration_noise_matches_its_restatement(act_dim, step):
    """rng_state mode: actions equal the explicit-noise mode fed with the NumPy restatement of the kernel's stream."""
    from safe_marl_amd.nets import fused_actor_forward
    agent, obs, hid = _agent_and_inputs(rows_b=13, act_dim=act_dim)
    seed = 0x1234567890ABCDE
    state = torch.tensor([seed, step], dtype=torch.int64, device="cuda")
    with torch.no_grad():
        m1, h1, a1, e1 = fused_actor_forward(agent, obs, hid, 5, True, rng_state=state, std=0.7, low=0.0, high=1.0)
        z = torch.from_numpy(_noise_restatement(seed, step, 13 * 5, act_dim)).float().cuda()
        m2, h2, a2, e2 = fused_actor_forward(agent, obs, hid, 5, True, noise=z.view(13, 5, act_dim), std=0.7, low=0.0, high=1.0)
    assert torch.equal(m1, m2) and torch.equal(h1, h2)
    assert (a1 - a2).abs().max().item() < 2e-5 and (e1 - e2).abs().max().item() < 2e-5
    assert int(state[1].item()) == step                       # the actor call itself does not advance the stream


def test_in_kernel_noise_is_standard_normal_and_moves_with_the_step():
    from safe_marl_amd.nets import fused_actor_forward
    agent, obs, hid = _agent_and_inputs(rows_b=8192, act_dim=4)
    with torch.no_grad():
        for p in agent.parameters():
            p.zero_()                                          # means = 0: action = tanh(std * z)
        outs = []
        for step in (0, 1):
            state = torch.tensor([99, step], dtype=torch.int64, device="cuda")
            _, _, act, _ = fused_actor_forward(agent, obs, hid, 5, True, rng_state=state, std=1.0, low=0.0, high=1.0)
            outs.append(torch.atanh(act.double().clamp(-1 + 1e-12, 1 - 1e-12)))
    z = outs[0]
    assert abs(z.mean().item()) < 0.01 and abs(z.var().item() - 1.0) < 0.02
    assert abs((z[:, 0] * z[:, 1]).mean().item()) < 0.01 and abs((z[:-1, 0] * z[1:, 0]).mean().item()) < 0.01
    assert abs((outs[0] * outs[1]).mean().item()) < 0.01       # consecutive steps are independent draws


@pytest.mark.parametrize("b", [4096, 1001, 7])
def test_rollout_size_kernel_equals_the_32_row_kernel_per_row(b):
    """The five-tiles-per-CU kernel (round 3: wavefronts 0-3 one 16-row tile each, wavefronts 4-7 a fifth tile shared by
    output units through LDS) against the 32-row kernel: every output unit is one MFMA chain over the inputs in the same
    order in both, whichever wavefront computes it — the same products, so the results agree to the last few ulps (the
    two kernels differ in how fp32 LayerNorm sums are grouped), and the kernel's result does not depend on the tile a row
    falls into: a batch permuted by whole rows gives the permuted result bit for bit."""
    from safe_marl_amd.nets import fused_actor_forward
    agent = _agent(144, 5, 4)
    g = torch.Generator(device="cuda").manual_seed(3)
    obs = torch.randn(b, 5, 144, device="cuda", generator=g)
    hid = torch.randn(b, 5, 64, device="cuda", generator=g)
    noise = torch.randn(b, 5, 4, device="cuda", generator=g)
    o16 = fused_actor_forward(agent, obs, hid, 5, True, noise=noise, variant=2)
    o32 = fused_actor_forward(agent, obs, hid, 5, True, noise=noise, variant=3)
    for x, y in zip(o16, o32):
        assert x.shape == y.shape and (x - y).abs().max().item() < 2e-6 * max(1.0, y.abs().max().item())
    # rows land in other tiles (full-tile wavefronts vs the cooperating four) after a rotation by 3 samples = 15 rows
    rot = lambda t: torch.roll(t, 3, 0).contiguous()
    r16 = fused_actor_forward(agent, rot(obs), rot(hid), 5, True, noise=rot(noise), variant=2)
    for x, y in zip(o16, r16):
        assert torch.equal(rot(x.view(b, 5, -1)), y.view(b, 5, -1))


@pytest.mark.parametrize("variant,n_envs", [(0, 64), (1, 64), (3, 64), (0, 4096), (3, 8192), (2, 1001)])
def test_observations_read_in_place_from_the_environment_history(variant, n_envs):
    """FlexActorArgs.obs_pushed (include/flexnet.h): the policy kernels read the stacked observation IN PLACE from the
    environment's mirror ring (flexenv_obs_source) — same means and hidden states, bit for bit, as from the materialised
    [N, n, 144] copy (flexenv_obs_view), with the environments at different positions of their rings (a staggered start:
    part of the batch restarts mid-run) and after the ring has wrapped."""
    from safe_marl_amd.flex_env import VecFlexProvisionEnv
    from safe_marl_amd.nets import fused_actor_forward
    from safe_marl_amd.network import create_network
    from safe_marl_amd.series import make_synthetic_series
    net = create_network()
    env = VecFlexProvisionEnv({}, n_envs, net=net, series=make_synthetic_series(net, n_days=20), seed=3, warm_start=True)
    agent = _agent(144, 5, 4)
    g = torch.Generator(device="cuda").manual_seed(5)
    env.reset()
    hid = torch.randn(n_envs, 5, 64, device="cuda", generator=g)
    mask = (torch.arange(n_envs, device="cuda") % 3 == 0).to(torch.uint8)
    for t in range(40):
        if t == 11:
            env.reset(mask=mask, want_obs=True)                 # a third of the batch starts a new episode: rings out of phase
        env.step(0.5 + 0.5 * torch.rand(n_envs, 5, 4, device="cuda", generator=g), obs_rows=True)
        if t in (0, 5, 12, 22, 23, 39):
            stacked = env.obs_view().clone()
            want = fused_actor_forward(agent, stacked, hid, 5, True, variant=variant)
            got = fused_actor_forward(agent, stacked, hid, 5, True, variant=variant, obs_source=env.obs_source())
            assert want is not None and got is not None
            assert torch.equal(want[0], got[0]) and torch.equal(want[1], got[1]), t
    pushed = env.peek("STEPS")
    assert int(pushed.min()) != int(pushed.max())               # the batch really was out of phase
