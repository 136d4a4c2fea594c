#!/usr/bin/env python3
"""Kernel times against batch rows (how the forward / backward kernels fill the 256 CUs)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jsrl-corl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch

import iql
import synth
from hip_helpers import to_torch_batch

S, A = int(os.environ.get("S", 17)), int(os.environ.get("A", 6))
BATCHES = [int(x) for x in os.environ.get("BATCHES", "32,64,128,192,224,256,288,320,384,512").split(",")]
qf, vf, actor = iql.TwinQ(S, A).cuda(), iql.ValueFunction(S).cuda(), iql.GaussianPolicy(S, A, 1.0).cuda()
tr = iql.ImplicitQLearning(1.0, actor, torch.optim.Adam(actor.parameters(), lr=3e-4), qf,
                           torch.optim.Adam(qf.parameters(), lr=3e-4), vf,
                           torch.optim.Adam(vf.parameters(), lr=3e-4), max_steps=1000000, device="cuda")
for B in BATCHES:
    d = synth.synth_transitions(B, S, A, seed=1)
    tb = to_torch_batch({"s": d["observations"], "a": d["actions"], "r": d["rewards"], "ns": d["next_observations"],
                         "d": d["terminals"]})
    tr.train(tb)
    if os.environ.get("PRECISION"):
        tr.set_precision(os.environ["PRECISION"])
    t = [tr.time_kernel(tb, w, 300) for w in (0, 1, 2, 3)]
    n_rt, n_chunk = (B + 31) // 32, (B + 255) // 256
    print(f"S={S} A={A} B={B:4d}: fwd {t[0]:6.2f} ({7 * n_rt * 4:4d} blocks)  bwd {t[1]:6.2f} ({4 * (32 * n_chunk + 4 * n_rt):4d} blocks)  "
          f"update {t[2]:5.2f}  all3 {t[3]:6.2f} us", flush=True)
