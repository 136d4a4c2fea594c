#!/usr/bin/env python3
"""Kernel times at BASELINE configs[4] geometry per GPU (obs=39, act=28, 1024 rows per GPU), fp32 and bf16 operands."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jsrl-corl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch

import iql
import synth
from hip_helpers import to_torch_batch

S, A = 39, 28
for B in (256, 1024, 8192):
    qf, vf, actor = iql.TwinQ(S, A).cuda(), iql.ValueFunction(S).cuda(), iql.GaussianPolicy(S, A, 1.0).cuda()
    tr = iql.ImplicitQLearning(1.0, actor, torch.optim.Adam(actor.parameters(), lr=3e-4), qf,
                               torch.optim.Adam(qf.parameters(), lr=3e-4), vf,
                               torch.optim.Adam(vf.parameters(), lr=3e-4), iql_tau=0.8, max_steps=1000000, device="cuda")
    d = synth.synth_transitions(B, S, A, seed=1)
    tb = to_torch_batch({"s": d["observations"], "a": d["actions"], "r": d["rewards"], "ns": d["next_observations"],
                         "d": d["terminals"]})
    tr.train(tb)
    for mode in ("f32", "bf16"):
        tr.set_precision(mode)
        t = [tr.time_kernel(tb, w, 200) for w in (0, 1, 2, 3)]
        print(f"S={S} A={A} B={B} {mode:4s}: fwd {t[0]:6.2f}  bwd {t[1]:6.2f}  update {t[2]:5.2f}  all3 {t[3]:6.2f} us/step "
              f"-> {B / t[3]:.1f} rows/us", flush=True)
    del tr
