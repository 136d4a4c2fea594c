#!/usr/bin/env python3
"""Where the fixed cost of a train_steps call goes: Python time in front of the library call, the library's own host
timestamps (IQLHIP_TRACE=1), host return, completion.  n = 20 is the driver's bench command."""
import os, sys, time
os.environ["IQLHIP_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jsrl-corl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import __graft_entry__ as ge
ge.build()
import iql, synth
import iqlhip_binding as hb
from hip_helpers import build_hip_trainer
S, A, B, N = 17, 6, 256, 1_000_000
data = synth.synth_transitions(N, S, A, seed=0)
buf = iql.ReplayBuffer(S, A, N, "cuda")
buf.load_d4rl_dataset(data)
params = synth.synth_params(S, A, seed=1)
tr = build_hip_trainer(params, S, A, True, {"iql_tau": .7, "beta": 3., "discount": .99, "tau": .005}, {"v": 3e-4, "q": 3e-4, "pi": 3e-4}, 1_000_000)
tr.prepare_train_steps(buf, B)
tr.train_steps(buf, 200, B, return_losses=False); torch.cuda.synchronize()
real = hb.lib().iqlhip_train_steps
t_in = [0.0]
class Spy:
    def __call__(self, *a):
        t_in[0] = time.perf_counter()
        return real(*a)
import ctypes as C
# how the wait ends: spinning on a host-mapped flag (no HIP call) vs torch.cuda.synchronize()
for n in (20, 20, 20, 64):
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.train_steps(buf, n, B, return_losses=False)
        sp = C.c_double(0)
        hb.check(hb.lib().iqlhip_debug_drain_spin(tr._ctx, tr._stream(), C.byref(sp)))
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"n={n:3d}: drained (flag spin) after {(t1-t0)*1e6:7.1f} us; torch.cuda.synchronize() returned {(t2-t1)*1e6:5.1f} us later", flush=True)
for n in (2, 4, 16, 20, 20, 20, 64):
    for rep in range(3):
        torch.cuda.synchronize()
        hb.lib().iqlhip_train_steps = Spy()
        t0 = time.perf_counter()
        tr.train_steps(buf, n, B, return_losses=False)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        hb.lib().iqlhip_train_steps = real
        print(f"n={n:3d}: python before the library call {(t_in[0]-t0)*1e6:6.1f} us | host returned {(t1-t0)*1e6:6.1f} | done {(t2-t0)*1e6:7.1f} us = {(t2-t0)*1e6/n:6.2f} us/step", flush=True)
