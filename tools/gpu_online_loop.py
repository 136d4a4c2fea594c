#!/usr/bin/env python3
"""The online / JSRL iteration of BASELINE configs[2] (obs=29, act=8, batch 256, 10 k-row ring) as the reference's
loop body runs it (algorithms/finetune/iql.py:725-778, jsrl_w_iql.py:512-554): act -> env.step -> add_transition ->
sample -> train, against a pure-numpy stand-in environment (no gym / mujoco in this image).  Reports iterations/s and
the share of each call."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jsrl-corl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import iql


class ToyEnv:
    """Linear-Gaussian dynamics with antmaze-shaped observations / actions; reward = -1 until a goal ball is hit."""

    def __init__(self, S, A, seed=0):
        self.rng = np.random.default_rng(seed)
        self.S, self.A = S, A
        self.M = self.rng.standard_normal((A, S)).astype(np.float32) * 0.1
        self.s = None
        self.t = 0

    def reset(self):
        self.s = self.rng.standard_normal(self.S).astype(np.float32)
        self.t = 0
        return self.s

    def step(self, a):
        self.s = (0.95 * self.s + a @ self.M + 0.05 * self.rng.standard_normal(self.S)).astype(np.float32)
        self.t += 1
        goal = bool(np.linalg.norm(self.s[:2]) < 0.05)
        done = goal or self.t >= 700
        return self.s, (0.0 if goal else -1.0), done, {}


def main(iters=3000, S=29, A=8, B=256, ring=10_000, fused=False):
    dev = "cuda"
    torch.manual_seed(0)
    np.random.seed(0)
    qf, vf, actor = iql.TwinQ(S, A).to(dev), iql.ValueFunction(S).to(dev), iql.GaussianPolicy(S, A, 1.0).to(dev)
    tr = iql.ImplicitQLearning(1.0, actor, torch.optim.Adam(actor.parameters(), lr=3e-4), qf,
                               torch.optim.Adam(qf.parameters(), lr=3e-4), vf, torch.optim.Adam(vf.parameters(), lr=3e-4),
                               iql_tau=0.9, beta=10.0, max_steps=1_000_000, device=dev)
    buf = iql.ReplayBuffer(S, A, ring, dev)
    env = ToyEnv(S, A)
    state = env.reset()
    for _ in range(B):                      # fill enough rows to sample from
        a = np.random.uniform(-1, 1, A).astype(np.float32)
        ns, r, d, _ = env.step(a)
        buf.add_transition(state, a, r, ns, d)
        state = env.reset() if d else ns
    t = {"act": 0.0, "env": 0.0, "add": 0.0, "sample": 0.0, "train": 0.0}
    next_a = None
    for it in range(iters + 200):
        if it == 200:
            torch.cuda.synchronize()
            t = {k: 0.0 for k in t}
            t0 = time.perf_counter()
        c0 = time.perf_counter()
        a = next_a if (fused and next_a is not None) else actor.act(state, dev)
        c1 = time.perf_counter()
        ns, r, d, _ = env.step(a)
        c2 = time.perf_counter()
        if fused:      # add_transition + sample + train as ONE library call (ImplicitQLearning.online_step)
            c3 = c4 = c2
            if d:
                log, next_a = tr.online_step(buf, state, a, r, ns, d, B), None
            else:          # the next iteration's act(ns) rides in the same call (same updated policy, one sync)
                log, next_a = tr.online_step(buf, state, a, r, ns, d, B, act_next=ns)
        else:
            buf.add_transition(state, a, r, ns, d)
            c3 = time.perf_counter()
            batch = buf.sample(B)
            batch = [b.to(dev) for b in batch]
            c4 = time.perf_counter()
            log = tr.train(batch)
        c5 = time.perf_counter()
        state = env.reset() if d else ns
        for k, v in zip(t, (c1 - c0, c2 - c1, c3 - c2, c4 - c3, c5 - c4)):
            t[k] += v
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert all(np.isfinite(v) for v in log.values())
    print(f"online loop{' [fused online_step]' if fused else ''} (S={S}, A={A}, B={B}, ring {ring}): {iters / dt:.0f} iterations/s, {dt / iters * 1e6:.1f} us each; "
          + ", ".join(f"{k} {v / iters * 1e6:.1f} us" for k, v in t.items()))
    return iters / dt


if __name__ == "__main__":
    main()
    main(fused=True)
