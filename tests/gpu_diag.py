#!/usr/bin/env python3
"""Stage-by-stage comparison of the HIP step against the oracle on the GPU box.
Prints the max error of every intermediate so a wrong kernel stage is visible
in one run (test infrastructure: it uses the oracle, so it lives under tests/).
Usage: python tests/gpu_diag.py [case ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jsrl-corl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

from helpers import load_golden, single_step_inputs, sub
from hip_helpers import (build_hip_trainer, head_values, read_moments, read_params, to_torch_batch,
                         unflatten_grads)
from oracle import iql_oracle as O


def err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(1e-30, float(np.max(np.abs(b)))))


def run(name):
    z, meta = load_golden(name)
    params, batch, hyper = single_step_inputs(meta)
    S, A, B = meta["S"], meta["A"], meta["B"]
    tr = build_hip_trainer(params, S, A, meta["gaussian"], hyper, meta["lrs"], meta["max_steps"])
    tb = to_torch_batch(batch)
    info = O.iql_losses_and_grads(params, batch, hyper)
    flat = tr.flat_gradient(tb)
    hv = head_values(tr, params, B)
    print(f"== {name}: S={S} A={A} B={B} gaussian={meta['gaussian']}")
    MB = tr._max_batch
    h0 = tr.debug_read("h0").reshape(4, MB, 256)[:, :B]
    h1 = tr.debug_read("h1").reshape(4, MB, 256)[:, :B]
    for i, n in enumerate(("vf", "q1", "q2", "pi")):
        print(f"  h0[{n}] {err(h0[i], info['acts'][n][0]):.2e}  h1[{n}] {err(h1[i], info['acts'][n][1]):.2e}")
    print(f"  next_v {err(hv['next_v'], info['next_v']):.2e} v {err(hv['v'], info['v']):.2e} "
          f"tq {err(np.minimum(hv['qt1'], hv['qt2']), info['target_q']):.2e} q1 {err(hv['q1'], info['q1']):.2e} "
          f"q2 {err(hv['q2'], info['q2']):.2e} mu {err(np.tanh(hv['pre']), info['mu']):.2e}")
    grads, lw = unflatten_grads(tr, flat)
    print(f"  losses hip {lw} oracle {[float(info['value_loss']), float(info['q_loss']), float(info['actor_loss'])]} "
          f"ref {z['losses']}")
    for n, t in grads.items():
        print("  grad", n, " ".join(f"{k}:{err(g, info['grads'][n][k]):.1e}" for k, g in t.items()))
    log = tr.train(tb)
    print("  train() losses", log)
    newp, newo, _ = O.iql_step(params, O.new_opt_state(params), batch, hyper, meta["lrs"])
    gp = read_params(tr)
    gm = read_moments(tr)
    for n, t in gp.items():
        print("  param", n, " ".join(f"{k}:{np.max(np.abs(p - newp[n][k])):.1e}" for k, p in t.items()))
    for n, t in gm["m"].items():
        print("  m    ", n, " ".join(f"{k}:{err(p, newo['m'][n][k]):.1e}" for k, p in t.items()))
    for n, t in gm["v"].items():
        print("  v    ", n, " ".join(f"{k}:{err(p, newo['v'][n][k]):.1e}" for k, p in t.items()))
    del tr


if __name__ == "__main__":
    cases = sys.argv[1:] or ["g1_S17A6_gauss_b3", "g1_S29A8_det_b10", "g1_S39A28_gauss_b3", "g1_ragged_B100"]
    for c in cases:
        run(c)
