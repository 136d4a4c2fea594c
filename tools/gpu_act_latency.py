#!/usr/bin/env python3
"""Latency of actor.act(state, "cuda") — the per-env-step call of the online / JSRL loops:
library path (pack -> policy forward -> finish, pinned staging) vs the same module's PyTorch forward
(what the reference does: torch.tensor(state) -> 3 addmm/relu/tanh -> clamp -> .cpu().numpy())."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jsrl-corl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import iql
import iqlhip_networks as nets

S, A = int(os.environ.get("S", 17)), int(os.environ.get("A", 6))
qf, vf, actor = iql.TwinQ(S, A).cuda(), iql.ValueFunction(S).cuda(), iql.GaussianPolicy(S, A, 1.0).cuda()
tr = iql.ImplicitQLearning(1.0, actor, torch.optim.Adam(actor.parameters(), lr=3e-4), qf,
                           torch.optim.Adam(qf.parameters(), lr=3e-4), vf, torch.optim.Adam(vf.parameters(), lr=3e-4),
                           max_steps=1000000, device="cuda")
rng = np.random.default_rng(0)
states = rng.standard_normal((2000, S)).astype(np.float32)


def bench(label, n=2000):
    for i in range(200):
        actor.act(states[i], "cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        actor.act(states[i % len(states)], "cuda")
    dt = (time.perf_counter() - t0) / n
    print(f"{label:34s} {dt * 1e6:8.1f} us/call")
    return dt


for mode, fn in (("eval (mean)", actor.eval), ("train (sample)", actor.train)):
    fn()
    t_hip = bench(f"act() library path, {mode}")
    ref = nets._ACTOR_OWNERS.pop(actor)
    t_torch = bench(f"act() PyTorch path, {mode}")
    nets._ACTOR_OWNERS[actor] = ref
    print(f"   speed-up {t_torch / t_hip:.2f}x")
actor.eval()
x = torch.from_numpy(rng.standard_normal((4096, S)).astype(np.float32)).cuda()
for n in (256, 4096):
    for _ in range(20):
        tr.actor_forward(x[:n])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        tr.actor_forward(x[:n])
    torch.cuda.synchronize()
    t_hip = (time.perf_counter() - t0) / 200
    with torch.no_grad():
        for _ in range(20):
            torch.clamp(actor(x[:n]).mean, -1, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            torch.clamp(actor(x[:n]).mean, -1, 1)
        torch.cuda.synchronize()
    t_torch = (time.perf_counter() - t0) / 200
    print(f"batched forward n={n}: library {t_hip * 1e6:.1f} us, PyTorch {t_torch * 1e6:.1f} us")
