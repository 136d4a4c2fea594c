#!/usr/bin/env python3
"""Wall time of consecutive synchronised train_steps calls of given lengths (first-use effects of a call length):
gpu_call_seq.py 5 20 20 20 ...   (IQLHIP_TRACE=1 adds the library's host-side stamps)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jsrl-corl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import __graft_entry__ as ge
ge.build()
import iql, synth
from hip_helpers import build_hip_trainer
seq = [int(x) for x in sys.argv[1:]] or [5, 20, 20, 20, 20]
S, A, B, N = 17, 6, 256, 1_000_000
data = synth.synth_transitions(N, S, A, seed=0)
buf = iql.ReplayBuffer(S, A, N, "cuda")
buf.load_d4rl_dataset(data)
params = synth.synth_params(S, A, seed=1)
tr = build_hip_trainer(params, S, A, True, {"iql_tau": .7, "beta": 3., "discount": .99, "tau": .005}, {"v": 3e-4, "q": 3e-4, "pi": 3e-4}, 1_000_000)
tr.prepare_train_steps(buf, B)
for n in seq:
    torch.cuda.synchronize()
    t = time.perf_counter()
    tr.train_steps(buf, n, B, seed=1234, return_losses=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"n={n:3d}: host returned {(t1-t)*1e6:6.1f} us, done {(t2-t)*1e6:7.1f} us = {(t2-t)*1e6/n:6.2f} us/step", flush=True)
