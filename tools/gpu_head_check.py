#!/usr/bin/env python3
"""Forward heads of one step against float64 (error in units of 1e-8) — debugging aid. IQLHIP_LIB selects the library."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jsrl-corl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np

import synth
from helpers import batch_from, load_golden
from hip_helpers import build_hip_trainer, head_values, to_torch_batch

name = os.environ.get("CASE", "g2_freerun_S29A8_det")
z, meta = load_golden(name)
S, A = meta["S"], meta["A"]
params = synth.synth_params(S, A, seed=meta["seed"], gaussian=meta["gaussian"])
data = synth.synth_transitions(meta["N"], S, A, seed=2000 + meta["seed"])
tr = build_hip_trainer(params, S, A, meta["gaussian"], dict(meta["hyper"]), meta["lrs"], meta["max_steps"], device="cuda:0")
b = batch_from(data, z["indices"][0])
B = len(b["r"])
tr.train(to_torch_batch(b, "cuda:0"))
hv = head_values(tr, params, B)


def mlp64(t, x):
    h0 = np.maximum(x @ t["w0"].astype(np.float64).T + t["b0"], 0)
    h1 = np.maximum(h0 @ t["w1"].astype(np.float64).T + t["b1"], 0)
    return h1 @ t["w2"].astype(np.float64).T + t["b2"]


sa = np.concatenate([b["s"], b["a"]], 1).astype(np.float64)
want = {"next_v": mlp64(params["vf"], b["ns"].astype(np.float64))[:, 0], "v": mlp64(params["vf"], b["s"].astype(np.float64))[:, 0],
        "qt1": mlp64(params["qt1"], sa)[:, 0], "qt2": mlp64(params["qt2"], sa)[:, 0],
        "q1": mlp64(params["q1"], sa)[:, 0], "q2": mlp64(params["q2"], sa)[:, 0]}
for k, w in want.items():
    e = np.abs(hv[k].astype(np.float64) - w)
    print(f"{k:7s} abs err max {e.max():.2e} rms {np.sqrt((e * e).mean()):.2e}  argmax row {int(e.argmax())}")
pre = mlp64(params["pi"], b["s"].astype(np.float64))
e = np.abs(hv["pre"].astype(np.float64) - pre)
print(f"pi pre  abs err max {e.max():.2e} rms {np.sqrt((e * e).mean()):.2e}")
