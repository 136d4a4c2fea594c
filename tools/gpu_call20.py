#!/usr/bin/env python3
"""Median / min wall time of the driver's timed region (synchronize; train_steps(n); synchronize) over many repetitions
in one process.  usage: gpu_call20.py [n=20] [reps=300]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jsrl-corl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import __graft_entry__ as ge
ge.build()
import iql, synth
from hip_helpers import build_hip_trainer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
S, A, B, N = 17, 6, 256, 1_000_000
data = synth.synth_transitions(N, S, A, seed=0)
buf = iql.ReplayBuffer(S, A, N, "cuda")
buf.load_d4rl_dataset(data)
params = synth.synth_params(S, A, seed=1)
tr = build_hip_trainer(params, S, A, True, {"iql_tau": .7, "beta": 3., "discount": .99, "tau": .005}, {"v": 3e-4, "q": 3e-4, "pi": 3e-4}, 1_000_000)
tr.prepare_train_steps(buf, B)
tr.train_steps(buf, 5, B, seed=1234, return_losses=False); torch.cuda.synchronize()
ts = []
for r in range(reps):
    torch.cuda.synchronize()
    t = time.perf_counter()
    tr.train_steps(buf, n, B, seed=1234, return_losses=False)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t) * 1e6)
ts = np.array(ts)
first = ts[0]
ts.sort()
print(f"n={n}: first {first:.1f} us, median {np.median(ts):.1f}, p10 {ts[len(ts)//10]:.1f}, min {ts[0]:.1f} us  -> {n/np.median(ts)*1e6:.0f} steps/s at the median"
      f"  [{' '.join(k+'='+v for k,v in os.environ.items() if k.startswith('IQLHIP_'))}]", flush=True)
