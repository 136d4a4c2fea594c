#!/usr/bin/env python3
"""Per-step time of the data-parallel step's LOCAL part on one GPU (1-rank group): plain step vs the RCCL exchange
(flatten + ncclAllReduce + update, all inside the chunk graph) vs the P2P exchange (backward writes into the exchange
block, update reads it back with system-scope loads; no flatten, no flag kernel at world 1).  What a multi-GPU step adds
on top is the flag handshake (one 64-thread kernel) and the xGMI reads of the peers' blocks."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jsrl-corl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch, torch.distributed as dist
import __graft_entry__ as ge
ge.build()
import iql, synth
from hip_helpers import build_hip_trainer
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
dist.init_process_group("gloo", rank=0, world_size=1)
S, A, B, N = 17, 6, 256, 1_000_000
buf = iql.ReplayBuffer(S, A, N, "cuda"); buf.load_d4rl_dataset(synth.synth_transitions(N, S, A, seed=0))
hyper = {"iql_tau": .7, "beta": 3., "discount": .99, "tau": .005}; lrs = {"v": 3e-4, "q": 3e-4, "pi": 3e-4}
for mode in (None, "rccl", "p2p"):
    tr = build_hip_trainer(synth.synth_params(S, A, seed=1), S, A, True, hyper, lrs, 1_000_000)
    if mode:
        tr.enable_data_parallel(exchange=mode)
    tr.prepare_train_steps(buf, B)
    tr.train_steps(buf, 2048, B, return_losses=False); torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); tr.train_steps(buf, 8192, B, return_losses=False); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 8192 * 1e6)
    print(f"exchange={mode}: {best:.2f} us/step (world 1)", flush=True)
dist.destroy_process_group()
