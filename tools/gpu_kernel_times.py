#!/usr/bin/env python3
"""Per-kernel time of the IQL step, each kernel launched back to back in isolation."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jsrl-corl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch

import iql
import synth
from hip_helpers import to_torch_batch

for (S, A, B) in ((17, 6, 256), (29, 8, 256), (39, 28, 256), (17, 6, 2048)):
    qf, vf, actor = iql.TwinQ(S, A).cuda(), iql.ValueFunction(S).cuda(), iql.GaussianPolicy(S, A, 1.0).cuda()
    tr = iql.ImplicitQLearning(1.0, actor, torch.optim.Adam(actor.parameters(), lr=3e-4), qf,
                               torch.optim.Adam(qf.parameters(), lr=3e-4), vf,
                               torch.optim.Adam(vf.parameters(), lr=3e-4), max_steps=1000000, device="cuda")
    d = synth.synth_transitions(B, S, A, seed=1)
    tb = to_torch_batch({"s": d["observations"], "a": d["actions"], "r": d["rewards"], "ns": d["next_observations"],
                         "d": d["terminals"]})
    tr.train(tb)
    t = [tr.time_kernel(tb, w, 300) for w in (0, 1, 2, 3)]
    print(f"S={S} A={A} B={B}: fwd {t[0]:.2f} us  bwd {t[1]:.2f} us  update {t[2]:.2f} us  all3 {t[3]:.2f} us/step", flush=True)
    tr.set_precision("bf16")
    t = [tr.time_kernel(tb, w, 300) for w in (0, 1, 2, 3)]
    print(f"   bf16 operands     : fwd {t[0]:.2f} us  bwd {t[1]:.2f} us  update {t[2]:.2f} us  all3 {t[3]:.2f} us/step", flush=True)
    del tr
