#!/usr/bin/env python3
"""Dump per-step flat gradients and parameters of a free-running fixture to an .npz (A/B debugging aid).
usage: IQLHIP_LIB=... python tools/gpu_step_dump.py out.npz"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jsrl-corl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np

import synth
from helpers import batch_from, load_golden
from hip_helpers import build_hip_trainer, to_torch_batch

name = os.environ.get("CASE", "g2_freerun_S29A8_det")
z, meta = load_golden(name)
S, A = meta["S"], meta["A"]
params = synth.synth_params(S, A, seed=meta["seed"], gaussian=meta["gaussian"])
data = synth.synth_transitions(meta["N"], S, A, seed=2000 + meta["seed"])
tr = build_hip_trainer(params, S, A, meta["gaussian"], dict(meta["hyper"]), meta["lrs"], meta["max_steps"], device="cuda:0")
out = {}
for k in range(meta["n_steps"]):
    tb = to_torch_batch(batch_from(data, z["indices"][k]), "cuda:0")
    out[f"g{k}"] = tr.flat_gradient(tb)
    tr.train(tb)
    out[f"p{k}"] = tr._params_arena.detach().cpu().numpy()
np.savez(sys.argv[1], **out)
L = tr._layout
print("layout q2:", {f: getattr(L.net[2], f) for f in ("w0", "b0", "w1", "b1", "w2", "b2", "k_in")})
