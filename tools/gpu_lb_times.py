#!/usr/bin/env python3
"""Kernel times of the bf16 path at BASELINE configs[4] dims (obs=39, act=28): tools/gpu_lb_times.py [B ...]
Environment switches of the library (read when the context is created) select the variant:
IQLHIP_LB=0 (small-batch kernels at every size), IQLHIP_LB_NBB / _CPB / _NBI (block geometry)."""
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
sizes = [int(x) for x in sys.argv[1:]] or [1024, 8192]
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("IQLHIP_"))
for B in sizes:
    qf, vf, actor = iql.TwinQ(S, A).cuda(), iql.ValueFunction(S).cuda(), iql.GaussianPolicy(S, A, 1.0).cuda()
    tr = iql.ImplicitQLearning(1.0, actor, torch.optim.Adam(actor.parameters(), lr=3e-4), qf,
                               torch.optim.Adam(qf.parameters(), lr=3e-4), vf,
                               torch.optim.Adam(vf.parameters(), lr=3e-4), iql_tau=0.8, max_steps=1000000, device="cuda")
    d = synth.synth_transitions(B, S, A, seed=1)
    tb = to_torch_batch({"s": d["observations"], "a": d["actions"], "r": d["rewards"], "ns": d["next_observations"],
                         "d": d["terminals"]})
    tr.train(tb)
    tr.set_precision("bf16")
    t = [tr.time_kernel(tb, w, 200) for w in (0, 1, 2, 3)]
    print(f"[{tag}] S={S} A={A} B={B} bf16: fwd {t[0]:6.2f}  bwd {t[1]:6.2f}  update {t[2]:5.2f}  all3 {t[3]:6.2f} us/step",
          flush=True)
    del tr
