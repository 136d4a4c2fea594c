#!/usr/bin/env python3
"""Wall time of train_steps(n) for several n (host-return time and time to completion): shows that the per-step cost
does not depend on n (graph chunks + direct steps) and what the fixed cost of a call is."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jsrl-corl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import __graft_entry__ as ge
ge.build()
import iql, synth
from hip_helpers import build_hip_trainer
S, A, B, N = 17, 6, 256, 1_000_000
data = synth.synth_transitions(N, S, A, seed=0)
buf = iql.ReplayBuffer(S, A, N, "cuda")
buf.load_d4rl_dataset(data)
params = synth.synth_params(S, A, seed=1)
tr = build_hip_trainer(params, S, A, True, {"iql_tau": .7, "beta": 3., "discount": .99, "tau": .005}, {"v": 3e-4, "q": 3e-4, "pi": 3e-4}, 1_000_000)
tr.prepare_train_steps(buf, B)
for n in (20, 20, 64, 64):      # first launches of the prepared (never launched) 16- and 64-step chunk graphs
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.train_steps(buf, n, B, return_losses=False); torch.cuda.synchronize()
    print(f"first-use n={n}: {(time.perf_counter() - t0) * 1e6:.1f} us", flush=True)
tr.train_steps(buf, 200, B, return_losses=False); torch.cuda.synchronize()
for n in (1, 5, 20, 63, 64, 65, 128, 200, 1000, 1024, 5000):
    best = None
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.train_steps(buf, n, B, return_losses=False)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        cur = ((t2 - t0) * 1e6, (t1 - t0) * 1e6)
        best = cur if best is None or cur[0] < best[0] else best
    print(f"n={n:5d}: total {best[0]:9.1f} us = {best[0]/n:7.2f} us/step; host returned after {best[1]:8.1f} us", flush=True)
