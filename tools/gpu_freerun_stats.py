#!/usr/bin/env python3
"""Per-tensor deviation of the HIP path from the reference's parameters after the free-running fixtures
(tests/golden/g2_*): max / rms / fraction above 2e-6.  IQLHIP_LIB selects the library (A/B runs)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jsrl-corl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np

import synth
from helpers import FREERUN_CASES, batch_from, load_golden, sub
from hip_helpers import build_hip_trainer, read_params, to_torch_batch

for name in FREERUN_CASES:
    z, meta = load_golden(name)
    S, A = meta["S"], meta["A"]
    params = synth.synth_params(S, A, seed=meta["seed"], gaussian=meta["gaussian"])
    data = synth.synth_transitions(meta["N"], S, A, seed=2000 + meta["seed"])
    tr = build_hip_trainer(params, S, A, meta["gaussian"], dict(meta["hyper"]), meta["lrs"], meta["max_steps"], device="cuda:0")
    worst = 0.0
    for k in range(meta["n_steps"]):
        log = tr.train(to_torch_batch(batch_from(data, z["indices"][k]), "cuda:0"))
        got = np.array([log["value_loss"], log["q_loss"], log["actor_loss"]])
        worst = max(worst, float(np.max(np.abs(got - z["losses"][k]) / np.abs(z["losses"][k]))))
    print(f"{name}: worst loss rel err over {meta['n_steps']} steps {worst:.2e}")
    for net, tensors in read_params(tr).items():
        for t, p in tensors.items():
            want = z[f"param.{net}.{t}"]
            d = np.abs(sub(p, meta["stride"]).reshape(want.shape).astype(np.float64) - want)
            print(f"   {net:4s}.{t:8s} n={d.size:6d} max {d.max():.2e} rms {np.sqrt((d * d).mean()):.2e} frac>2e-6 {(d > 2e-6).mean():.4f}")
