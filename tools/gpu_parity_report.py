#!/usr/bin/env python3
"""Worst observed error of the HIP step per (reference fixture, tensor class), printed as a table
(-> profiles/r02_parity_margins.txt).  The tolerances in tests/ are set from this table; SURVEY §8d's targets are
losses rel 1e-5, gradients |dg|inf <= 1e-5*max(1,|g|inf), parameters abs 2e-6, Adam moments rel 1e-5, target abs 1e-7.

    python tools/gpu_parity_report.py > profiles/r02_parity_margins.txt        (on the GPU box)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jsrl-corl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import __graft_entry__ as ge

ge.build()
import synth
from helpers import SINGLE_STEP_CASES, load_golden, single_step_inputs, sub
from hip_helpers import build_hip_trainer, read_moments, read_params, to_torch_batch, unflatten_grads

CASES = SINGLE_STEP_CASES + ["g9_dropout_S39A28_gauss", "g9_dropout_S17A6_det", "g14_c5_B1024_dropout"]


def errs(z, meta, info, newp, newo):
    stride = meta["stride"]
    out = {}

    def put(cls, e, where):
        if cls not in out or e > out[cls][0]:
            out[cls] = (e, where)

    got = np.array([info["value_loss"], info["q_loss"], info["actor_loss"]], dtype=np.float64)
    want = z["losses"].astype(np.float64)
    put("loss rel", float(np.max(np.abs(got - want) / np.abs(want))), "losses")
    for net, tensors in info["grads"].items():
        for t, g in tensors.items():
            key = f"grad.{net}.{t}"
            if key not in z:
                continue
            w = z[key].astype(np.float64)
            d = float(np.max(np.abs(sub(g, stride).reshape(w.shape).astype(np.float64) - w)))
            put("grad |d|inf / |g|inf", d / max(float(np.max(np.abs(w))), 1e-30), key)
            put("grad |d|inf / max(1,|g|inf)  [SURVEY 1e-5]", d / max(1.0, float(np.max(np.abs(w)))), key)
    n_amp = 0
    for net, tensors in newp.items():
        for t, p in tensors.items():
            key = f"param.{net}.{t}"
            if key not in z:
                continue
            w = z[key].astype(np.float64)
            diff = np.abs(sub(p, stride).reshape(w.shape).astype(np.float64) - w)
            if net in ("qt1", "qt2"):
                put("target abs  [SURVEY 1e-7]", float(diff.max()), key)
                continue
            gkey = f"grad.{net}.{t}"
            if gkey in z:   # split off the elements whose first Adam step is ill-conditioned (|g| within noise of eps)
                g = np.abs(z[gkey].astype(np.float64)).reshape(diff.shape)
                dg = 2e-6 * max(float(g.max()), 1e-30)
                amp = 3e-4 * np.minimum(1.0, 1e-8 * dg / (g + 1e-8) ** 2) > 1e-7
                n_amp += int(amp.sum())
                if (~amp).any():
                    put("param abs (well-conditioned elements)  [SURVEY 2e-6]", float(diff[~amp].max()), key)
                if amp.any():
                    put("param abs (|g| ~ Adam eps, amplified)", float(diff[amp].max()), key)
            else:
                put("param abs (well-conditioned elements)  [SURVEY 2e-6]", float(diff.max()), key)
    for mv in ("m", "v"):
        for net, tensors in newo[mv].items():
            for t, a in tensors.items():
                key = f"{mv}.{net}.{t}"
                if key not in z:
                    continue
                w = z[key].astype(np.float64)
                d = float(np.max(np.abs(sub(a, stride).reshape(w.shape).astype(np.float64) - w)))
                put(f"Adam {mv} |d|inf / |{mv}|inf  [SURVEY 1e-5]", d / max(float(np.max(np.abs(w))), 1e-30), key)
    out["(elements with amplified Adam step)"] = (float(n_amp), "")
    return out


def main():
    print(f"# HIP step vs reference fixtures, worst observed error per tensor class ({torch.cuda.get_device_name(0)})")
    overall = {}
    for name in CASES:
        z, meta = load_golden(name)
        params, batch, hyper = single_step_inputs(meta)
        drop = meta.get("dropout", 0.0)
        tr = build_hip_trainer(params, meta["S"], meta["A"], meta["gaussian"], hyper, meta["lrs"], meta["max_steps"],
                               dropout=drop)
        if drop:
            k0, k1 = synth.synth_dropout_keep(meta["B"], drop, seed=meta["seed"])
            tr.inject_dropout_masks(k0, k1)
        tb = to_torch_batch(batch)
        grads, lw = unflatten_grads(tr, tr.flat_gradient(tb))
        log = tr.train(tb)
        info = {"value_loss": log["value_loss"], "q_loss": log["q_loss"], "actor_loss": log["actor_loss"], "grads": grads}
        e = errs(z, meta, info, read_params(tr), read_moments(tr))
        print(f"\n{name}  (S={meta['S']} A={meta['A']} B={meta['B']} {'gauss' if meta['gaussian'] else 'det'})")
        for cls, (v, where) in sorted(e.items()):
            print(f"  {cls:58s} {v:10.3e}  {where}")
            if cls not in overall or v > overall[cls][0]:
                overall[cls] = (v, name + ":" + where)
    print("\n== worst over all fixtures ==")
    for cls, (v, where) in sorted(overall.items()):
        print(f"  {cls:58s} {v:10.3e}  {where}")


if __name__ == "__main__":
    main()
