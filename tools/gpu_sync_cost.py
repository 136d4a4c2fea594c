#!/usr/bin/env python3
"""What does the synchronisation at the end of a train_steps call cost the host, once the GPU has drained?
(the GPU-side end is observed by spinning on a host-mapped flag, no HIP call)"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jsrl-corl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import __graft_entry__ as ge
ge.build()
import iql, synth
import iqlhip_binding as hb
from hip_helpers import build_hip_trainer
S, A, B, N = 17, 6, 256, 200_000
data = synth.synth_transitions(N, S, A, seed=0)
buf = iql.ReplayBuffer(S, A, N, "cuda")
buf.load_d4rl_dataset(data)
params = synth.synth_params(S, A, seed=1)
tr = build_hip_trainer(params, S, A, True, {"iql_tau": .7, "beta": 3., "discount": .99, "tau": .005}, {"v": 3e-4, "q": 3e-4, "pi": 3e-4}, 1_000_000)
tr.prepare_train_steps(buf, B)
tr.train_steps(buf, 200, B, return_losses=False); torch.cuda.synchronize()
lib = hb.lib()
def drain():
    sp = C.c_double(0)
    hb.check(lib.iqlhip_debug_drain_spin(tr._ctx, tr._stream(), C.byref(sp)))
def us(f):
    t = time.perf_counter(); f(); return (time.perf_counter() - t) * 1e6
ev = torch.cuda.Event()
def med(xs): return sorted(xs)[len(xs) // 2]
res = {}
for name in ("idle: device sync", "work+drain: device sync", "work+drain: stream sync", "work+drain: stream query", "work+event+drain: device sync",
             "work+event+drain: event sync then device sync", "work (no drain): device sync total", "work+event (no drain): device sync total",
             "work+drain+query: device sync"):
    xs = []
    for rep in range(9):
        torch.cuda.synchronize()
        if name.startswith("idle"):
            xs.append(us(torch.cuda.synchronize)); continue
        t0 = time.perf_counter()
        tr.train_steps(buf, 20, B, return_losses=False)
        if "event" in name: ev.record()
        if "no drain" in name:
            torch.cuda.synchronize(); xs.append((time.perf_counter() - t0) * 1e6); continue
        drain()
        if name.endswith("stream sync"): xs.append(us(lambda: lib.iqlhip_stream_synchronize(tr._stream())))
        elif name.endswith("stream query"): xs.append(us(lambda: torch.cuda.current_stream().query()))
        elif "event sync then" in name:
            a = us(ev.synchronize); b = us(torch.cuda.synchronize); xs.append(a + b)
        elif "drain+query" in name:
            torch.cuda.current_stream().query(); xs.append(us(torch.cuda.synchronize))
        else: xs.append(us(torch.cuda.synchronize))
    print(f"{name:50s}: median {med(xs):7.1f} us  (min {min(xs):7.1f})", flush=True)
