#!/usr/bin/env python3
"""Trajectory of consecutive synchronised 20-step calls (done / host-returned, us), then the same right behind a
sustained 2 000-step call (GPU clocks up, host path cold for ~40 ms) and behind 200 one-step calls (host path warm)."""
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
tr.train_steps(buf, 5, B, seed=1234, return_losses=False)
def calls(n, reps):
    out = []
    for r in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        tr.train_steps(buf, n, B, seed=1234, return_losses=False)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        out.append(((t2 - t) * 1e6, (t1 - t) * 1e6))
    return out
def show(tag, o):
    print(tag)
    print("  done:     " + " ".join(f"{a:.0f}" for a, _ in o))
    print("  returned: " + " ".join(f"{b:.0f}" for _, b in o), flush=True)
show("right behind prepare + a 5-step call: 60 calls of 20", calls(20, 60))
time.sleep(0.5)
show("after 0.5 s of sleep: 20 calls of 20", calls(20, 20))
tr.train_steps(buf, 2000, B, seed=1234, return_losses=False)
show("right behind a 2 000-step call: 20 calls of 20", calls(20, 20))
time.sleep(0.5)
calls(1, 200)
show("after 0.5 s of sleep + 200 one-step calls: 20 calls of 20", calls(20, 20))
