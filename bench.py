#!/usr/bin/env python3
"""bench.py — IQL gradient-steps/sec at batch=256 on N MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch: draw 256 row indices
(uniform with replacement), gather the rows from the HBM-resident replay buffer,
7 MLP forwards, 3 losses, backward, Adam x3, Polyak.  Inputs are synthetic
D4RL-shaped rows resident in HBM before the timed region starts.

N = 1: BASELINE.json configs[1] (obs=17, act=6, 1M rows, batch=256): K steps replayed
       as hipGraph chunks (ImplicitQLearning.train_steps).
N > 1: configs[3]-style data parallelism at fixed per-GPU batch 256 (global batch 256*N,
       weak scaling): per-rank device index draw, forward+backward, RCCL all-reduce of the
       flat gradient, fused Adam/Polyak on every rank (ImplicitQLearning.train_on_buffer).
       value = batch-256 gradient computations per second summed over ranks.

Prints ONE JSON line on rank 0 (contract in the round prompt) with `roofline` (the
backward kernel, the step's dominant launch, against the fp32-MFMA peak) and, at N = 1,
`cpu_baseline` (oracle/iql_torch_port.py — a PyTorch-CPU port of the reference step —
timed on the host cores for a bounded ~20 s sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "jsrl-corl_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_* dense peak
HID = 256


def flops(S, A, B):
    """Algorithmic FLOPs per launch: 2*M*N*K per GEMM, biases/elementwise excluded."""
    def w(k0, d):
        return k0 * HID + HID * HID + HID * d
    wV, wQ, wP = w(S, 1), w(S + A, 1), w(S, A)
    fwd = 2 * B * (2 * wV + 4 * wQ + wP)
    bwd = 2 * B * ((2 * wV - S * HID) + 2 * (2 * wQ - (S + A) * HID) + (2 * wP - S * HID))
    return fwd, bwd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=1000)
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--state-dim", type=int, default=17)
    ap.add_argument("--action-dim", type=int, default=6)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--precision", choices=("f32", "bf16"), default="f32",
                    help="f32 = the parity path (headline); bf16 = bf16 operands / fp32 accumulate in the 256-deep products")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    args = ap.parse_args()

    import numpy as np
    import torch

    import __graft_entry__ as ge
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    ndev = max(1, torch.cuda.device_count())
    dev_index = local_rank % ndev          # (rehearsals put several ranks on one GPU; the driver gives one GPU per rank)
    backend = os.environ.get("IQLHIP_DIST_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm; "gloo" only for rehearsals
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    # one rank compiles (a no-op when the prebuilt library is current); the others wait before loading it
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    ge._paths()
    import iql
    import synth

    S, A, B = args.state_dim, args.action_dim, args.batch
    dev = f"cuda:{dev_index}"
    torch.cuda.set_device(dev_index)

    # ---- synthetic HBM-resident buffer + nets (seed 0)
    data = synth.synth_transitions(args.rows, S, A, seed=0)
    buf = iql.ReplayBuffer(S, A, args.rows, dev)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        buf.load_d4rl_dataset(data)
    del data
    torch.manual_seed(0)
    qf, vf, actor = iql.TwinQ(S, A).to(dev), iql.ValueFunction(S).to(dev), iql.GaussianPolicy(S, A, 1.0).to(dev)
    tr = iql.ImplicitQLearning(
        max_action=1.0, actor=actor, actor_optimizer=torch.optim.Adam(actor.parameters(), lr=3e-4),
        q_network=qf, q_optimizer=torch.optim.Adam(qf.parameters(), lr=3e-4),
        v_network=vf, v_optimizer=torch.optim.Adam(vf.parameters(), lr=3e-4),
        iql_tau=0.7, beta=3.0, max_steps=1_000_000, discount=0.99, tau=0.005, device=dev)
    if args.precision == "bf16":
        tr.set_precision("bf16")
    force_dp = os.environ.get("IQLHIP_BENCH_FORCE_DP") == "1"   # diagnostic: 1-rank process group, DP code path
    if force_dp and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29731")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", dev_index))
        tr._dp_world, tr._dp_group = 2, None      # take the split path; the collective runs over 1 rank
        tr._dp_world_scale = 1
    if world > 1:
        tr.enable_data_parallel()

    def run(n):
        if world > 1 or force_dp:
            tr.train_steps_dp(buf, n, B, seed=1234)
        else:
            tr.train_steps(buf, n, B, seed=1234, return_losses=False)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    run(args.warmup)
    fence()
    t0 = time.perf_counter()
    run(args.steps)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # sanity: the run trained (losses finite)
    log = tr.train(buf.sample(B)) if world == 1 else tr.train_on_buffer(buf, B, seed=1, sync=True)
    assert all(np.isfinite(v) for v in log.values()), log

    # ---- roofline of the dominant kernel (backward): its average launch duration is measured live with
    # HIP events on the launch stream around back-to-back launches of that kernel on the bench batch
    # (iqlhip_debug_time_kernel; events around single ~10 us launches would add their own ~3 us).
    f_fwd, f_bwd = flops(S, A, B)
    roof = None
    if world == 1:
        batch = buf.sample(B)
        t_fwd = tr.time_kernel(batch, 0, 500)
        t_bwd = tr.time_kernel(batch, 1, 500)
        t_upd = tr.time_kernel(batch, 2, 500)
        ach = f_bwd / (t_bwd * 1e-6) / 1e12
        # HBM-side bytes per launch of that kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE and
        # --pmc WRITE_SIZE in separate runs of this bench, (2*FETCH_SIZE + WRITE_SIZE)*1024 per the gfx950
        # correction; tools/pmc_traffic.py).  Only valid for the default workload.
        traffic, mfma_util = None, None
        pmc = os.path.join(ROOT, "profiles", "r01_final_pmc_summary.json")
        if os.path.exists(pmc) and (S, A, B) == (17, 6, 256) and args.precision == "f32":
            with open(pmc) as fh:
                for name, rec in json.load(fh).items():
                    if "iql_bwd_kernel" in name:
                        traffic = rec.get("hbm_bytes_per_launch_corrected")
                        mfma_util = rec.get("mfma_util")      # SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz)
        roof = {"bound": "mfma", "kernel": "iql_bwd_kernel", "achieved": round(ach, 3), "peak": PEAK_F32_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                "traffic": None if traffic is None else round(traffic),
                "mfma_util_pmc": None if mfma_util is None else round(mfma_util, 4),
                "flops_per_launch": f_bwd, "avg_launch_us": round(t_bwd, 3),
                "kernel_us": {"iql_fwd_kernel": round(t_fwd, 3), "iql_bwd_kernel": round(t_bwd, 3),
                              "iql_update_kernel": round(t_upd, 3)},
                "step_flops": f_fwd + f_bwd,
                "step_frac_of_peak": round((f_fwd + f_bwd) * args.steps / dt / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)}

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    value = args.steps * world / dt
    out = {
        "metric": "IQL gradient-steps/sec at batch=256 (D4RL obs/act dims)",
        "value": round(value, 1),
        "unit": "steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 5),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.precision,
        "data": "synthetic",
        "config": {"workload": f"IQL step on synthetic buffer (obs={S}, act={A}, {args.rows} rows), batch={B} per GPU",
                   "global_batch": B * world, "parallelism": f"dp{world}" if world > 1 else "single",
                   "global_steps_per_s": round(args.steps / dt, 1)},
    }
    if roof:
        out["roofline"] = roof
    if world == 1 and not args.no_cpu_baseline:
        from oracle import iql_torch_port as port
        ncpu = os.cpu_count() or 1
        sps1, n1, el1 = port.time_cpu_steps(S, A, B, min(args.rows, 200_000), seconds_budget=args.cpu_seconds, threads=1)
        thr = min(ncpu, 16)
        spsN, nN, elN = port.time_cpu_steps(S, A, B, min(args.rows, 200_000), seconds_budget=args.cpu_seconds, threads=thr)
        best, cores = (sps1, 1) if sps1 >= spsN else (spsN, thr)
        out["cpu_baseline"] = {
            "value": round(best, 2), "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"PyTorch-CPU port of the reference step (oracle/iql_torch_port.py), same S/A/B, 200k-row buffer: "
                      f"{n1} steps in {el1:.1f}s @1 thread = {sps1:.1f}/s; {nN} steps in {elN:.1f}s @{thr} threads = {spsN:.1f}/s; "
                      f"torch {torch.__version__}, {ncpu} host cpus"}
        out["speedup_vs_cpu"] = round(value / best, 1)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
