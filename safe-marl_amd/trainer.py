"""PGTrainer with the reference's constructor and methods (utils/trainer.py:10-138):
``PGTrainer(args, model_cls, env, logger)``, ``.run(stat, episode)``, ``.logging``, ``.print_info``,
``.behaviour_net``, ``.replay_buffer``, ``.steps``, ``.episodes`` and the
``{policy,value}_replay_process`` hooks that ``Model.transition_update`` calls (model.py:48,50).

Differences by design: the replay buffer lives on the device and hands out tensor windows (no
``Transition(*zip(*batch))`` re-columnisation, trainer.py:68); with more than one rank the flattened
gradient bucket is all-reduced (mean) over RCCL BEFORE the grad-norm clip; with a
vectorised env the batch is ``batch_size * batch_scale`` transitions.
"""
from __future__ import annotations

import logging

import torch as th
from torch import optim

from . import dist as fdist
from .replay_buffer import TransReplayBuffer
from .util import get_grad_norm, normal_entropy

train_logger = logging.getLogger("TrainLogger")


class PGTrainer(object):
    def __init__(self, args, model, env, logger, batch_scale=None, replay_capacity=None):
        self.args = args
        self.env = env
        self.device = th.device("cuda" if th.cuda.is_available() and self.args.cuda else "cpu")
        self.logger = logger
        self.episodic = self.args.episodic
        if self.episodic:
            raise NotImplementedError("episodic replay is outside the MADDPG hot path (default.yaml:9)")
        needs_env = args.alg == "safemaddpg"                                     # trainer.py:17-28
        make = (lambda *a: model(self.args, self.env, *a)) if needs_env else (lambda *a: model(self.args, *a))
        if self.args.target:
            target_net = make().to(self.device)
            self.behaviour_net = make(target_net).to(self.device)
        else:
            self.behaviour_net = make().to(self.device)
        n_envs = getattr(env, "n_envs", 1)
        self.batch_scale = batch_scale if batch_scale is not None else max(1, n_envs // 4)
        if n_envs == 1:
            self.batch_scale = 1 if batch_scale is None else batch_scale
        if self.args.replay:
            cap = replay_capacity if replay_capacity is not None else int(self.args.replay_buffer_size) * max(1, min(n_envs, 64))
            self.replay_buffer = TransReplayBuffer(cap, device=self.device)     # trainer.py:29-33
        self.policy_optimizer = optim.RMSprop(self.behaviour_net.policy_dicts.parameters(), lr=args.policy_lrate,
                                              alpha=0.99, eps=1e-5)            # trainer.py:34
        self.value_optimizer = optim.RMSprop(self.behaviour_net.value_dicts.parameters(), lr=args.value_lrate,
                                             alpha=0.99, eps=1e-5)             # trainer.py:35
        self.init_action = th.zeros(1, self.args.agent_num, self.args.action_dim).to(self.device)
        self.steps = 0
        self.episodes = 0
        self.entr = self.args.entr
        self.world = fdist.world_size()
        if self.world > 1:      # identical replicas: rank 0's weights everywhere
            fdist.broadcast_module(self.behaviour_net)

    def effective_batch_size(self):
        return self.args.batch_size * self.batch_scale

    def get_loss(self, batch, need="both"):
        return self.behaviour_net.get_loss(batch, need=need)                     # trainer.py:43-45

    def _sample(self):
        return self.replay_buffer.get_batch_tensors(self.effective_batch_size())   # trainer.py:67,72

    def policy_replay_process(self, stat):
        self.policy_transition_process(stat, self._sample())

    def value_replay_process(self, stat):
        self.value_transition_process(stat, self._sample())

    def _finish(self, params, optimizer):
        if self.world > 1:
            fdist.allreduce_grads(params)                                        # mean over ranks, one flat bucket
        grad_norm = get_grad_norm(self.args, params)                             # util.py:159-161, after the all-reduce
        optimizer.step()
        return grad_norm

    def policy_transition_process(self, stat, trans):
        """trainer.py:81-97 (continuous branch), incl. the constant entropy term of trainer.py:47-57."""
        policy_loss, _, logits = self.get_loss(trans, need="policy")
        means, log_stds = logits
        self.policy_optimizer.zero_grad()
        if self.entr > 0:
            entropy = normal_entropy(means, log_stds.exp())
            policy_loss = policy_loss - self.entr * entropy
            stat["mean_train_entropy"] = entropy.detach()
        policy_loss.backward()
        params = self.policy_optimizer.param_groups[0]["params"]
        norm = self._finish(params, self.policy_optimizer)
        stat["mean_train_policy_grad_norm"] = norm.detach()
        stat["mean_train_policy_loss"] = policy_loss.detach()

    def value_transition_process(self, stat, trans):
        """trainer.py:99-108."""
        _, value_loss, _ = self.get_loss(trans, need="value")
        self.value_optimizer.zero_grad()
        value_loss.backward()
        params = self.value_optimizer.param_groups[0]["params"]
        norm = self._finish(params, self.value_optimizer)
        stat["mean_train_value_grad_norm"] = norm.detach()
        stat["mean_train_value_loss"] = value_loss.detach()

    def run(self, stat, episode):
        """trainer.py:120-124."""
        self.behaviour_net.train_process(stat, self)
        if (episode % self.args.eval_freq == self.args.eval_freq - 1) or (episode == 0):
            self.behaviour_net.evaluation(stat, self)
        for k, v in list(stat.items()):          # device scalars -> floats, once per episode
            if isinstance(v, th.Tensor):
                stat[k] = float(v.item())

    def logging(self, stat):
        """trainer.py:126-130."""
        if self.logger is None:
            return
        for k, v in stat.items():
            self.logger.add_scalar("data/" + k, v, self.episodes)

    def print_info(self, stat):
        """trainer.py:132-138."""
        string = [f"\nEpisode: {self.episodes}"] + [f"{k}: {float(v):2.4f}" for k, v in stat.items()]
        train_logger.info("\n".join(string))
