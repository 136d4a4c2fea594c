#!/usr/bin/env python3
"""Phase breakdown of the large-batch bf16 kernels (iqlhip_lb_kernels.h) from in-kernel s_memtime stamps.
IQLHIP_LIB=jsrl-corl_amd/libiqlhip_stamps.so B=1024 python tools/gpu_lb_stamps.py   (library: tools/build_variant.sh
libiqlhip_stamps.so -DIQL_STAMPS).  Phases are those of a block's FIRST row tile / chunk; 'rest' = its other tiles."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jsrl-corl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import iql
import synth
from hip_helpers import to_torch_batch

S, A, B = int(os.environ.get("S", 39)), int(os.environ.get("A", 28)), int(os.environ.get("B", 1024))
qf, vf, actor = iql.TwinQ(S, A).cuda(), iql.ValueFunction(S).cuda(), iql.GaussianPolicy(S, A, 1.0).cuda()
tr = iql.ImplicitQLearning(1.0, actor, torch.optim.Adam(actor.parameters(), lr=3e-4), qf,
                           torch.optim.Adam(qf.parameters(), lr=3e-4), vf, torch.optim.Adam(vf.parameters(), lr=3e-4),
                           iql_tau=0.8, max_steps=1000000, device="cuda")
d = synth.synth_transitions(B, S, A, seed=1)
tb = to_torch_batch({"s": d["observations"], "a": d["actions"], "r": d["rewards"], "ns": d["next_observations"],
                     "d": d["terminals"]})
d0 = synth.synth_transitions(256, S, A, seed=2)      # (the context is created by a small fp32 step: the diagnostic build's
tr.train(to_torch_batch({"s": d0["observations"], "a": d0["actions"], "r": d0["rewards"], "ns": d0["next_observations"],
                         "d": d0["terminals"]}))      #  stamp buffer is sized for the launches this tool looks at)
tr.set_precision("bf16")
for it in range(int(os.environ.get("ITERS", 300))):
    tr.train(tb)
torch.cuda.synchronize()
st = tr.debug_read("stamps").view(np.uint64).reshape(4096, 16)

n_rt, n_chunk = (B + 31) // 32, (B + 255) // 256
even = (n_rt + 1) & ~1
nbi = min(even, int(os.environ.get("IQLHIP_LB_NBI", 32)))
nbb = min(even, int(os.environ.get("IQLHIP_LB_NBB", 32)))
cpb = int(os.environ.get("IQLHIP_LB_CPB", max(1, min(8, n_chunk // 2))))
n_cg = (n_chunk + cpb - 1) // cpb


def report(name, blk, labels):
    blk = blk[blk[:, 0] > 0]
    if len(blk) == 0:
        print(name, ": no blocks")
        return
    prev = 0
    print(f"{name}: {len(blk)} blocks")
    for i, lab in labels:
        ok = blk[:, i] > 0
        if ok.sum() == 0:
            continue
        dlt = blk[ok, i].astype(np.int64) - blk[ok, prev].astype(np.int64)
        print(f"   {lab:44s} median {np.median(dlt):8.0f}  p90 {np.percentile(dlt, 90):8.0f}  max {dlt.max():8d} cycles")
        prev = i
    tot = blk[:, 9].astype(np.int64) - blk[:, 0].astype(np.int64)
    t0, t1 = blk[:, 14].astype(np.int64), blk[:, 15].astype(np.int64)
    print(f"   block total median {np.median(tot):8.0f} max {tot.max():8d} cycles;  wall: start spread {(t0.max() - t0.min()) / 100:.2f} us, "
          f"duration median {np.median(t1 - t0) / 100:.2f} max {(t1 - t0).max() / 100:.2f} us, last end {(t1.max() - t0.min()) / 100:.2f} us after first start")


FWD = [(1, "persistent operand loads issued"), (2, "X tile -> LDS + barrier (waits for the loads)"), (3, "layer 0 + H0 tile"), (4, "barrier"),
       (5, "H0 save, layer-1 reads + 64 MFMAs"), (6, "epilogue, heads"), (7, "barrier"), (8, "H1 save (+ policy loss terms)"), (9, "rest of the block's row tiles")]
ids = np.arange(8 * nbi)
inst = np.where(((ids >> 3) & 1) == 0, np.array([1, 4, 5, 6])[ids & 3], np.array([0, 2, 3, 7])[ids & 3])
NAMES = ("V(s')", "V(s)", "Qt1", "Qt2", "Q1", "Q2", "PI")
for i in (0, 1, 4, 6):
    report(f"fwd {NAMES[i]}", st[ids[inst == i]], FWD)
report("fwd all", st[ids[inst != 7]], FWD)
cpb = int(os.environ.get("IQLHIP_LB_CPB", max(1, min(8, n_chunk // 4))))
n_cg = (n_chunk + cpb - 1) // cpb
csplit = even <= 32 and os.environ.get("IQLHIP_LB_CSPLIT", "1") != "0" and "IQLHIP_LB_NBB" not in os.environ
ids = np.arange(8 * nbb if csplit else 8 * (nbb // 2))      # (column split: two blocks per row tile, half = bit 0 of blockIdx >> 3)
net = ids & 3
LR = [(1, "first tile's loads + persistent operand loads issued"), (2, "H1 tile -> LDS, dy / dY (waits for the loads)"), (3, "barrier"), (4, "dH1 tile + barrier"),
      (5, "dH1 store, dH0 reads + 64 MFMAs"), (6, "mask, dH0 tile, next tile's loads + barrier"), (7, "dH0 store"), (8, "rest of the row tiles"), (9, "block sums -> slab")]
for n, nm in enumerate(("V", "Q1", "Q2", "PI")):
    if csplit:
        for ch in (0, 1):
            report(f"bwd rows {nm} column half {ch}", st[2048 + ids[(net == n) & (((ids >> 3) & 1) == ch)]], LR)
    else:
        report(f"bwd rows {nm}", st[2048 + ids[net == n]], LR)
report("bwd rows all", st[2048 + ids], LR)
ng = 8 * ((28 * n_cg + 1) // 2)
ids = np.arange(ng)
local = (ids >> 3) * 2 + ((ids & 7) >> 2)
job = local % 28
gs = st[3072 + ids]


def greport(name, blk):
    blk = blk[(blk[:, 0] > 0) & (blk[:, 3] > 0)]
    if len(blk) == 0:
        print(name, ": no blocks")
        return
    print(f"{name}: {len(blk)} blocks")
    for i, j, lab in ((0, 1, "stage 0: loads -> LDS + barrier"), (1, 4, "stage 0: next loads issued, tr reads + 8 MFMAs"), (4, 5, "stage 1 -> LDS (waits for its loads)"), (5, 6, "barrier"),
                      (6, 2, "remaining stages"), (2, 3, "tile -> slab")):
        ok = (blk[:, i] > 0) & (blk[:, j] > 0)
        if ok.sum():
            dlt = blk[ok, j].astype(np.int64) - blk[ok, i].astype(np.int64)
            print(f"   {lab:44s} median {np.median(dlt):8.0f}  p90 {np.percentile(dlt, 90):8.0f}  max {dlt.max():8d} cycles")
    tot = blk[:, 3].astype(np.int64) - blk[:, 0].astype(np.int64)
    t0, t1 = blk[:, 14].astype(np.int64), blk[:, 15].astype(np.int64)
    print(f"   block total median {np.median(tot):8.0f} max {tot.max():8d} cycles;  wall: start spread {(t0.max() - t0.min()) / 100:.2f} us, "
          f"duration median {np.median(t1 - t0) / 100:.2f} max {(t1 - t0).max() / 100:.2f} us, last end {(t1.max() - t0.min()) / 100:.2f} us after first start")


greport("bwd gemm dW1 tiles", gs[job < 16])
greport("bwd gemm dW0 tiles", gs[(job >= 16) & (job < 24)])
greport("bwd gemm dW2 tiles", gs[job >= 24])
greport("bwd gemm all", gs)
