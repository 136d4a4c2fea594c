#!/usr/bin/env python3
"""First timed call vs later ones: wall time (synchronize .. synchronize) beside the GPU-side time between two events
recorded on the trainer's stream around the call (from the moment the GPU reaches the first event to the end of the work)."""
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
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
tr.train_steps(buf, 5, B, seed=1234, return_losses=False)
rows = []
for i in range(40):
    e0, e1 = evs[i]
    torch.cuda.synchronize()
    t = time.perf_counter()
    e0.record()
    tr.train_steps(buf, 20, B, seed=1234, return_losses=False)
    t1 = time.perf_counter()
    e1.record()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    rows.append(((t2 - t) * 1e6, (t1 - t) * 1e6, e0.elapsed_time(e1) * 1e3))
print("call: wall us | host returned | GPU e0->e1 us | wall - GPU")
for i, (w, r, g) in enumerate(rows):
    if i < 12 or i >= 36:
        print(f"{i:3d}: {w:7.1f} | {r:6.1f} | {g:7.1f} | {w-g:6.1f}")
