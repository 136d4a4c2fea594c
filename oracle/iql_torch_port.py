"""ORACLE / CPU BASELINE — test infrastructure, NOT product code.

PyTorch-CPU port of the reference's IQL step, kept as close as possible to the
reference's *cost structure* (eager aten ops, autograd, three torch.optim.Adam
steps, per-tensor Polyak copy_, three .item() syncs, numpy index draw + five
advanced-index gathers), because bench.py times it on the GPU box's host cores
as `cpu_baseline` (kind "port").  The reference's own files cannot travel to the
GPU box, so this port stands in for them there.

Follows /root/reference/algorithms/finetune/iql.py:
  sample   :171-178        train :542-563        _update_v :482-495
  _update_q :497-515       _update_policy :517-540       soft_update :72-74
It is checked against the reference-generated goldens by
tests/test_oracle_golden.py::test_torch_port_matches_reference.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import it.
"""
from __future__ import annotations

import copy
import os
import sys
from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F
from torch.optim.lr_scheduler import CosineAnnealingLR

_PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "jsrl-corl_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)
import iqlhip_networks as nets  # plain nn.Modules (parameter containers); no HIP involved on CPU


class CpuReplay:
    """Five separate fp32 tensors + numpy index draw, like the reference's buffer."""

    def __init__(self, data: Dict[str, np.ndarray]):
        self.s = torch.from_numpy(data["observations"])
        self.a = torch.from_numpy(data["actions"])
        self.r = torch.from_numpy(data["rewards"][:, None].copy())
        self.ns = torch.from_numpy(data["next_observations"])
        self.d = torch.from_numpy(data["terminals"][:, None].astype(np.float32))
        self.size = self.s.shape[0]

    def sample(self, batch_size: int) -> List[torch.Tensor]:
        idx = np.random.randint(0, self.size, size=batch_size)
        return [self.s[idx], self.a[idx], self.r[idx], self.ns[idx], self.d[idx]]


class CpuIQL:
    def __init__(self, S, A, params=None, gaussian=True, iql_tau=0.7, beta=3.0, discount=0.99, tau=0.005,
                 lrs=None, max_steps=1000000):
        lrs = lrs or {"v": 3e-4, "q": 3e-4, "pi": 3e-4}
        self.qf = nets.TwinQ(S, A)
        self.vf = nets.ValueFunction(S)
        self.actor = (nets.GaussianPolicy if gaussian else nets.DeterministicPolicy)(S, A, 1.0)
        if params is not None:
            self._load(params)
        self.q_target = copy.deepcopy(self.qf).requires_grad_(False)
        if params is not None:
            self._load_mlp(self.q_target.q1, params["qt1"])
            self._load_mlp(self.q_target.q2, params["qt2"])
        self.v_opt = torch.optim.Adam(self.vf.parameters(), lr=lrs["v"])
        self.q_opt = torch.optim.Adam(self.qf.parameters(), lr=lrs["q"])
        self.a_opt = torch.optim.Adam(self.actor.parameters(), lr=lrs["pi"])
        self.sched = CosineAnnealingLR(self.a_opt, max_steps) if max_steps is not None else None
        self.iql_tau, self.beta, self.discount, self.tau = iql_tau, beta, discount, tau
        self.total_it = 0

    @staticmethod
    def _load_mlp(mod, t):
        lin = nets.linear_layers(mod)
        with torch.no_grad():
            for i, (w, b) in enumerate((("w0", "b0"), ("w1", "b1"), ("w2", "b2"))):
                lin[i].weight.copy_(torch.from_numpy(t[w]))
                lin[i].bias.copy_(torch.from_numpy(t[b]))

    def _load(self, p):
        self._load_mlp(self.vf.v, p["vf"])
        self._load_mlp(self.qf.q1, p["q1"])
        self._load_mlp(self.qf.q2, p["q2"])
        self._load_mlp(self.actor.net, p["pi"])
        if "log_std" in p["pi"]:
            with torch.no_grad():
                self.actor.log_std.copy_(torch.from_numpy(p["pi"]["log_std"]))

    def train(self, batch) -> Dict[str, float]:
        self.total_it += 1
        obs, act, rew, nobs, done = batch
        log = {}
        with torch.no_grad():
            next_v = self.vf(nobs)
            target_q = self.q_target(obs, act)
        # value update
        v = self.vf(obs)
        adv = target_q - v
        v_loss = torch.mean(torch.abs(self.iql_tau - (adv < 0).float()) * adv ** 2)
        log["value_loss"] = v_loss.item()
        self.v_opt.zero_grad()
        v_loss.backward()
        self.v_opt.step()
        # q update
        rew = rew.squeeze(dim=-1)
        done = done.squeeze(dim=-1)
        targets = rew + (1.0 - done.float()) * self.discount * next_v.detach()
        qs = self.qf.both(obs, act)
        q_loss = sum(F.mse_loss(q, targets) for q in qs) / len(qs)
        log["q_loss"] = q_loss.item()
        self.q_opt.zero_grad()
        q_loss.backward()
        self.q_opt.step()
        for tp, sp in zip(self.q_target.parameters(), self.qf.parameters()):
            tp.data.copy_((1 - self.tau) * tp.data + self.tau * sp.data)
        # policy update
        exp_adv = torch.exp(self.beta * adv.detach()).clamp(max=100.0)
        out = self.actor(obs)
        if isinstance(out, torch.distributions.Distribution):
            bc = -out.log_prob(act).sum(-1, keepdim=False)
        else:
            bc = torch.sum((out - act) ** 2, dim=1)
        pi_loss = torch.mean(exp_adv * bc)
        log["actor_loss"] = pi_loss.item()
        self.a_opt.zero_grad()
        pi_loss.backward()
        self.a_opt.step()
        if self.sched is not None:
            self.sched.step()
        return log


def time_cpu_steps(S, A, B, n_rows, seconds_budget=15.0, threads=1, warmup=5, seed=0, max_steps=None):
    """steps/s of sample()+train() on `threads` host threads over a bounded sample: `warmup` untimed steps (themselves
    bounded by the budget), then steps until `max_steps` (>= 2 000 after 200 warm-up) or the budget."""
    import time

    import synth
    torch.set_num_threads(threads)
    data = synth.synth_transitions(n_rows, S, A, seed=seed)
    buf = CpuReplay(data)
    tr = CpuIQL(S, A, params=synth.synth_params(S, A, seed=seed))
    np.random.seed(seed)
    t_w = time.perf_counter()
    for _ in range(warmup):
        tr.train(buf.sample(B))
        if time.perf_counter() - t_w > seconds_budget:      # (a slow host: the warm-up may not eat the whole run)
            break
    n, t0 = 0, time.perf_counter()
    while True:
        tr.train(buf.sample(B))
        n += 1
        el = time.perf_counter() - t0
        if (el >= seconds_budget and n >= 5) or (max_steps is not None and n >= max_steps):   # (checked every step)
            break
    return n / el, n, el
