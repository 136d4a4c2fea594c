#!/usr/bin/env python3
"""N eager bf16 steps at configs[4] dims on a fixed batch (a profiling target): tools/gpu_lb_step.py B [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jsrl-corl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch

import iql
import synth
from hip_helpers import to_torch_batch

S, A = int(os.environ.get("LB_S", 39)), int(os.environ.get("LB_A", 28))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
qf, vf, actor = iql.TwinQ(S, A).cuda(), iql.ValueFunction(S).cuda(), iql.GaussianPolicy(S, A, 1.0).cuda()
tr = iql.ImplicitQLearning(1.0, actor, torch.optim.Adam(actor.parameters(), lr=3e-4), qf,
                           torch.optim.Adam(qf.parameters(), lr=3e-4), vf, torch.optim.Adam(vf.parameters(), lr=3e-4),
                           iql_tau=0.8, max_steps=1000000, device="cuda")
d = synth.synth_transitions(B, S, A, seed=1)
tb = to_torch_batch({"s": d["observations"], "a": d["actions"], "r": d["rewards"], "ns": d["next_observations"],
                     "d": d["terminals"]})
tr.train(tb)
tr.set_precision(os.environ.get("PRECISION", "bf16"))
for _ in range(n):
    log = tr.train(tb)
torch.cuda.synchronize()
print(B, log)
