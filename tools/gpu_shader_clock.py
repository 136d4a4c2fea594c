#!/usr/bin/env python3
"""Shader clock during the step kernels, from the diagnostic library's per-block stamps: s_memtime (shader cycles) and
s_memrealtime (100 MHz) at block entry and exit of the forward kernel — after a long run of graph-replayed steps and after
a run of eager steps.  IQLHIP_LIB=jsrl-corl_amd/libiqlhip_stamps.so python tools/gpu_shader_clock.py"""
import os, sys, contextlib, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jsrl-corl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import iql, synth
S, A, B, N = 17, 6, 256, 200_000
buf = iql.ReplayBuffer(S, A, N, "cuda")
with contextlib.redirect_stdout(io.StringIO()):
    buf.load_d4rl_dataset(synth.synth_transitions(N, S, A, seed=0))
qf, vf, actor = iql.TwinQ(S, A).cuda(), iql.ValueFunction(S).cuda(), iql.GaussianPolicy(S, A, 1.0).cuda()
tr = iql.ImplicitQLearning(1.0, actor, torch.optim.Adam(actor.parameters(), lr=3e-4), qf,
                           torch.optim.Adam(qf.parameters(), lr=3e-4), vf, torch.optim.Adam(vf.parameters(), lr=3e-4),
                           max_steps=1000000, device="cuda")
def clock(tag):
    torch.cuda.synchronize()
    st = tr.debug_read("stamps").view(np.uint64).reshape(4096, 16)[:256]
    ok = (st[:, 0] > 0) & (st[:, 4] > st[:, 0]) & (st[:, 15] > st[:, 14])
    cyc = (st[ok, 4] - st[ok, 0]).astype(np.float64)
    ns = (st[ok, 15] - st[ok, 14]).astype(np.float64) * 10.0
    print(f"{tag}: forward blocks {int(ok.sum())}: {np.median(cyc):.0f} cycles in {np.median(ns) / 1e3:.2f} us -> {np.median(cyc / ns):.2f} GHz (p10 {np.percentile(cyc / ns, 10):.2f}, p90 {np.percentile(cyc / ns, 90):.2f})", flush=True)
tr.prepare_train_steps(buf, B)
for n in (20, 200, 2000, 20000):
    tr.train_steps(buf, n, B, return_losses=False)
    clock(f"after {n:6d} graph steps")
for n in (50, 3000):
    for _ in range(n):
        tr.train_on_buffer(buf, B, seed=7)
    clock(f"after {n:6d} eager steps")
tr.train_steps(buf, 20000, B, return_losses=False)
clock("after  20000 graph steps")
