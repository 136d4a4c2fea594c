#!/usr/bin/env python3
"""Why is the first timed train_steps(20) after prepare + 5 warm-up steps ~25 us slower than later ones?
Variants: nothing / a burst of throw-away GPU work right before (clock ramp?) / repeated calls."""
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
def fresh():
    tr = build_hip_trainer(params, S, A, True, {"iql_tau": .7, "beta": 3., "discount": .99, "tau": .005}, {"v": 3e-4, "q": 3e-4, "pi": 3e-4}, 1_000_000)
    tr.prepare_train_steps(buf, B)
    return tr
def timed(tr, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.train_steps(buf, n, B, seed=1234, return_losses=False); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6
x = torch.randn(4096, 4096, device="cuda")
WARM = int(os.environ.get("WARM", "5"))
for variant in ("plain", "burst 20 ms of matmul before the warm-up", "idle 50 ms before the warm-up", "burst then idle 2 ms"):
    tr = fresh()
    if variant.startswith("burst"):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.02:
            y = x @ x
        torch.cuda.synchronize()
    if "idle 50" in variant: time.sleep(0.05)
    if "idle 2 ms" in variant: time.sleep(0.002)
    tr.train_steps(buf, WARM, B, seed=1234, return_losses=False)
    ts = [timed(tr, 20) for _ in range(8)]
    print(f"{variant:45s}: " + " ".join(f"{t:6.1f}" for t in ts), flush=True)
