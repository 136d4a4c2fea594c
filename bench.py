#!/usr/bin/env python3
"""bench.py — IQL gradient-steps/sec at batch=256 on N MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch: draw 256 row indices
(uniform with replacement), gather the rows from the HBM-resident replay buffer,
7 MLP forwards, 3 losses, backward, Adam x3, Polyak.  Inputs are synthetic
D4RL-shaped rows resident in HBM before the timed region starts.

N = 1: BASELINE.json configs[1] (obs=17, act=6, 1M rows, batch=256).
N > 1: BASELINE.json configs[3] (obs=17, act=6, 10M-row buffer replicated per GPU, global batch 256*N sharded
       over the ranks — weak scaling at 256 rows per GPU): per-rank device index draw, forward+backward, exchange
       of the flat gradient over xGMI, fused Adam/Polyak on every rank.  The exchange is either RCCL's all-reduce or
       the library's direct peer-read exchange; both run inside the captured step graphs, --exchange picks one,
       the default ("auto") times both in the warm-up and keeps the faster one that left the replicas identical.
       value = batch-256 gradient computations per second summed over ranks.
       Launched without torch.distributed.run (WORLD_SIZE unset), `--gpus N` spawns the N rank processes itself.

The timed region never contains a graph capture or instantiation for ANY --steps/--warmup: the library composes a
run from its first 2 or 4 steps launched directly plus replays of fixed chunk graphs (64 / 16 / 4 / 2 / 1 steps; captured,
instantiated and rehearsed by prepare_train_steps before the warm-up).

Prints ONE JSON line on rank 0 (contract in the round prompt) with `roofline` (the backward kernel, the step's
dominant launch, against the fp32-MFMA peak) and, at N = 1, `cpu_baseline` (oracle/iql_torch_port.py — a PyTorch-CPU
port of the reference step — timed on the host cores for a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "jsrl-corl_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_* dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0
HID = 256
ROWS_1GPU = 1_000_000          # BASELINE.json configs[1]
ROWS_DP = 10_000_000           # BASELINE.json configs[3]
# rocprofv3 PMC passes of this bench command, collected by tools/profile_round.sh and committed (the newest round's file)
PMC_SUMMARIES = [os.path.join(ROOT, "profiles", f) for f in ("r04_pmc_summary.json", "r03_pmc_summary.json", "r02_pmc_summary.json")]
# ... and of `bench.py --state-dim 39 --action-dim 28 --batch 1024 --precision bf16` (BASELINE configs[4]'s per-GPU share),
# tools/profile_round.sh with BENCH_ARGS set
PMC_SUMMARIES_C5 = [os.path.join(ROOT, "profiles", f) for f in ("r04_pmc_config5.json",)]
LB_MIN_ROWS = 512              # bf16 batches above this take the large-batch kernels (csrc/iqlhip_lb_kernels.h)


def flops(S, A, B):
    """Algorithmic FLOPs per launch: 2*M*N*K per GEMM, biases/elementwise excluded."""
    def w(k0, d):
        return k0 * HID + HID * HID + HID * d
    wV, wQ, wP = w(S, 1), w(S + A, 1), w(S, A)
    fwd = 2 * B * (2 * wV + 4 * wQ + wP)
    bwd = 2 * B * ((2 * wV - S * HID) + 2 * (2 * wQ - (S + A) * HID) + (2 * wP - S * HID))
    return fwd, bwd


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=1000)
    ap.add_argument("--rows", type=int, default=0, help="buffer rows (default: 1M at --gpus 1, 10M per GPU above)")
    ap.add_argument("--state-dim", type=int, default=17)
    ap.add_argument("--action-dim", type=int, default=6)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--precision", choices=("f32", "bf16"), default="f32",
                    help="f32 = the parity path (headline); bf16 = bf16 operands / fp32 accumulate in the layer and weight-gradient products")
    ap.add_argument("--exchange", choices=("auto", "rccl", "p2p"), default="auto",
                    help="N > 1: gradient exchange (auto = time both in the warm-up, keep the faster valid one)")
    ap.add_argument("--repeats", type=int, default=0,
                    help="timed repeats of --steps, the MEDIAN is reported (default: 5 when --steps >= 1000, else 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=14.0,
                    help="budget per thread setting of the cpu_baseline leg (after 200 warm-up steps; it stops at 2 000 steps)")
    ap.add_argument("--timeout", type=float, default=540.0, help="self-launched ranks: overall wall-clock bound in seconds")
    ap.add_argument("--dry-run-launch", action="store_true",
                    help="print the rank fan-out this invocation would start (JSON) and exit; touches no GPU")
    ap.add_argument("--master-port", type=int, default=0)
    return ap.parse_args(argv)


def launch_plan(args, argv):
    """The N child processes `--gpus N` starts when no launcher set WORLD_SIZE: one per GPU, rank = local rank = GPU
    index, rendezvous on 127.0.0.1.  Returned as data so that a CPU test can check it without starting anything."""
    port = args.master_port or (29500 + os.getpid() % 2000)
    child_argv = [a for a in argv if a != "--dry-run-launch"]
    plan = []
    for r in range(args.gpus):
        env = {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus),
               "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
        plan.append({"rank": r, "cmd": [sys.executable, os.path.abspath(__file__)] + child_argv, "env": env})
    return plan


def spawn_ranks(args, argv) -> int:
    """Parent of a self-launched multi-GPU run.  It has not touched the GPU (no torch import, no HIP call) and never
    will: the ranks are fresh child processes; rank 0's stdout (the JSON line) is passed through.  All children are
    polled together: the first one that exits non-zero (out of memory on the 10 M-row buffer, a failed attach) takes
    the others down at once instead of leaving them in a collective until its timeout; the whole run is bounded."""
    plan = launch_plan(args, argv)
    procs = []
    for item in plan:
        env = dict(os.environ)
        env.update(item["env"])
        out = None if item["rank"] == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen(item["cmd"], env=env, stdout=out))
    return wait_ranks(procs, args.timeout)


def wait_ranks(procs, timeout_s: float, poll_s: float = 0.05) -> int:
    """0 when every child exited 0; otherwise the first failure's code (124 on the overall timeout) after the rest has
    been terminated (then killed)."""
    deadline = time.monotonic() + timeout_s
    rc = 0
    while True:
        codes = [pr.poll() for pr in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad:
            rc = bad[0]
            break
        if all(c == 0 for c in codes):
            return 0
        if time.monotonic() > deadline:
            sys.stderr.write(f"bench.py: ranks still running after {timeout_s:.0f} s, terminating them\n")
            rc = 124
            break
        time.sleep(poll_s)
    for pr in procs:
        if pr.poll() is None:
            pr.terminate()
    t_kill = time.monotonic() + 5.0
    for pr in procs:
        while pr.poll() is None and time.monotonic() < t_kill:
            time.sleep(poll_s)
        if pr.poll() is None:
            pr.kill()
            pr.wait()
    return rc


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    env_world = os.environ.get("WORLD_SIZE")
    if args.dry_run_launch:
        plan = launch_plan(args, argv) if (env_world is None and args.gpus > 1) else []
        print(json.dumps({"gpus": args.gpus, "self_launch": bool(plan), "ranks": plan,
                          "rows": args.rows or (ROWS_1GPU if args.gpus == 1 else ROWS_DP)}))
        return 0
    if env_world is None and args.gpus > 1:
        return spawn_ranks(args, argv)
    world = int(env_world or "1")
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch one rank per GPU "
                         f"(python bench.py --gpus N starts them itself)\n")
        return 2
    return run_rank(args, world)


def run_rank(args, world: int) -> int:
    import numpy as np
    import torch

    import __graft_entry__ as ge
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
    rows = args.rows or (ROWS_1GPU if world == 1 else ROWS_DP)
    dev = f"cuda:{dev_index}"
    torch.cuda.set_device(dev_index)

    # ---- synthetic HBM-resident buffer + nets (seed 0; every rank holds the same replicated buffer)
    # (rows are generated on the device, identically on every rank: Philox fill kernel with synth.py's distributions —
    #  at 10 M rows the host generator cost every rank ~30 s of numpy and a 1.7 GB upload)
    # wall-clock of the start-up phases (rank 0's view; every rank does the same work): the driver's 8-GPU run has a 600 s
    # limit, and none of this is in the timed region
    phases = {}
    t_ph = [time.perf_counter()]

    def phase(name):
        torch.cuda.synchronize()
        now = time.perf_counter()
        phases[name] = round(now - t_ph[0], 3)
        t_ph[0] = now

    buf = iql.ReplayBuffer(S, A, rows, dev)
    buf.fill_synthetic(rows, seed=0)
    phase("buffer_fill_s")
    torch.manual_seed(0)
    qf, vf, actor = iql.TwinQ(S, A).to(dev), iql.ValueFunction(S).to(dev), iql.GaussianPolicy(S, A, 1.0).to(dev)
    tr = iql.ImplicitQLearning(
        max_action=1.0, actor=actor, actor_optimizer=torch.optim.Adam(actor.parameters(), lr=3e-4),
        q_network=qf, q_optimizer=torch.optim.Adam(qf.parameters(), lr=3e-4),
        v_network=vf, v_optimizer=torch.optim.Adam(vf.parameters(), lr=3e-4),
        iql_tau=0.7, beta=3.0, max_steps=1_000_000, discount=0.99, tau=0.005, device=dev)
    tr.reserve_batch(B)
    if args.precision == "bf16":
        tr.set_precision("bf16")
    phase("trainer_setup_s")

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    eager_dp = [False]      # N > 1 with no in-library exchange available: one library step + torch all-reduce per step

    def run(n):
        if eager_dp[0]:
            for _ in range(n):
                tr.train_on_buffer(buf, B, seed=1234)
            return
        tr.train_steps(buf, n, B, seed=1234, return_losses=False)

    def timed(n):
        fence()
        t = time.perf_counter()
        run(n)
        fence()
        t = time.perf_counter() - t
        if world > 1:
            tt = torch.tensor([t], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t = float(tt.item())
        return t

    def replicas_equal() -> bool:
        """All ranks hold bit-identical parameters and no peer wait timed out."""
        p = tr._params_arena.view(torch.int32).to(torch.int64)
        sig = torch.stack([p.sum(), (p * torch.arange(1, p.numel() + 1, device=p.device)).sum()])
        lo, hi = sig.clone(), sig.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        bad = torch.tensor([0 if tr.exchange_status()["timed_out_step"] == 0 else 1], device=dev)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        return bool(torch.equal(lo, hi)) and int(bad.item()) == 0

    # ---- exchange (N > 1): attach, pre-capture, pick
    exchange, probe, why = None, {}, None
    warm = args.warmup
    if world > 1:
        want = args.exchange
        tr.enable_data_parallel(exchange={"auto": "both", "rccl": "rccl", "p2p": "p2p"}[want])
        phase("exchange_attach_s")
        modes = ["p2p", "rccl"] if want == "auto" else [want]
        if want == "auto":
            for m, attr in (("p2p", "_p2p_error"), ("rccl", "_rccl_error")):
                if getattr(tr, attr, None):              # that exchange could not be attached on some rank
                    modes.remove(m)
                    probe[m] = {"unavailable": getattr(tr, attr)}
        attached = list(modes)
        for m in modes:
            tr.select_exchange(m)
            tr.prepare_train_steps(buf, B)
            phase(f"prepare_{m}_s")

        def probe_mode(m, n):
            """n steps on exchange m: rate, and whether it left the replicas bit-identical with no wait timed out.  A
            probe that did not is undone: rank 0's state is broadcast again, the timeout word cleared."""
            tr.select_exchange(m)
            t = timed(n)
            if os.environ.get("IQLHIP_BENCH_BREAK_PROBE") == m and rank == world - 1:
                tr._params_arena[:4] += 1e-3      # test hook: this probe leaves the replicas REALLY diverged
            ok = replicas_equal()
            probe[m] = {"steps_per_s": round(n * world / t, 1), "replicas_equal": ok}
            if not ok:
                tr.resync_replicas()
                probe[m]["resynced"] = True
            return ok

        if want == "auto":
            # the warm-up steps are split between the attached exchanges and timed; the fastest one that kept the
            # replicas bit-identical runs the timed region; with none left: the eager torch.distributed exchange
            share = max(8, warm // 2) if len(modes) > 1 else 0
            if len(modes) > 1:
                ok = [m for m in modes if probe_mode(m, share)]
                warm = max(0, warm - share * len(modes))
                if ok:
                    exchange = max(ok, key=lambda m: probe[m]["steps_per_s"])
                    why = f"fastest of the exchanges that kept the replicas identical in a {share}-step probe each: {ok}"
            elif len(modes) == 1:
                if probe_mode(modes[0], max(8, min(warm, 64))):
                    exchange = modes[0]
                    why = "the only in-library exchange that could be attached (see exchange_probe) and it kept the replicas identical"
            if exchange is None:
                exchange = "torch"
                why = "no in-library exchange is usable here (see exchange_probe): eager torch.distributed all-reduce per step"
        else:
            exchange = want
            why = "selected with --exchange"
            if not probe_mode(want, max(8, min(warm, 64))):
                raise RuntimeError(f"--exchange {want}: the replicas diverged or a peer wait timed out in the probe: {probe}")
        if exchange == "torch":
            eager_dp[0] = True
            tr.select_exchange("torch")
        else:
            tr.select_exchange(exchange)
        phase("exchange_probe_s")
    else:
        tr.prepare_train_steps(buf, B)       # capture + instantiate + upload the chunk graph: never in the timed region
        phase("prepare_s")

    if warm > 0:
        run(warm)
    repeats = args.repeats if args.repeats > 0 else (5 if args.steps >= 1000 else 1)
    dts = [timed(args.steps) for _ in range(repeats)]      # EXACTLY --steps per timed region; the median region is reported
    dt = sorted(dts)[len(dts) // 2]

    # sanity: the run trained (losses finite), replicas still identical
    log = tr.train(buf.sample(B)) if world == 1 else tr.train_on_buffer(buf, B, seed=1, sync=True)
    assert all(np.isfinite(v) for v in log.values()), log
    if world > 1:
        assert replicas_equal(), "replicas diverged or a peer wait timed out"

    # ---- roofline of the dominant kernel (backward): its average launch duration is measured live with
    # HIP events on the launch stream around back-to-back launches of that kernel on the bench batch
    # (iqlhip_debug_time_kernel; events around single ~10 us launches would add their own ~3 us).
    f_fwd, f_bwd = flops(S, A, B)
    roof = None
    lb = args.precision == "bf16" and B > LB_MIN_ROWS and S + A + 1 <= 80 and os.environ.get("IQLHIP_LB", "1") != "0"
    if rank == 0:
        batch = buf.sample(B)
        t_fwd = tr.time_kernel(batch, 0, 500)
        t_bwd = tr.time_kernel(batch, 1, 500)
        t_upd = tr.time_kernel(batch, 2, 500)
        # the dominant kernel.  Small-batch kernels: the backward (one launch).  Large-batch bf16 path: the backward is two
        # launches (row kernel + row-contraction GEMM), the forward the longest single one
        kernel_name, t_dom, f_dom = "iql_bwd_kernel", t_bwd, f_bwd
        kernel_us = {"iql_fwd_kernel": round(t_fwd, 3), "iql_bwd_kernel": round(t_bwd, 3), "iql_update_kernel": round(t_upd, 3)}
        if lb:
            t_rows = tr.time_kernel(batch, 4, 500)
            t_gemm = tr.time_kernel(batch, 5, 500)
            heads = 3 + A                                     # head dims of V, Q1, Q2, pi
            f_rows = 2 * B * HID * (4 * HID + heads)          # dH0 = dH1 . W1 (x 4 nets), dH1 = dY . W2
            f_gemm = f_bwd - f_rows                           # dW1, dW0, dW2: every contraction over the batch rows
            kernel_us = {"iql_fwd_lb_kernel": round(t_fwd, 3), "iql_bwd_rows_kernel": round(t_rows, 3),
                         "iql_bwd_gemm_kernel": round(t_gemm, 3), "iql_update_kernel": round(t_upd, 3)}
            cands = [("iql_fwd_lb_kernel", t_fwd, f_fwd), ("iql_bwd_rows_kernel", t_rows, f_rows), ("iql_bwd_gemm_kernel", t_gemm, f_gemm)]
            kernel_name, t_dom, f_dom = max(cands, key=lambda c_: c_[1])
        ach = f_dom / (t_dom * 1e-6) / 1e12
        # HBM-side bytes per launch of that kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE and
        # --pmc WRITE_SIZE in separate runs of this bench, (2*FETCH_SIZE + WRITE_SIZE)*1024 per the gfx950
        # correction; tools/pmc_traffic.py).  Only valid for the default workload.
        traffic, mfma_util, pmc_src = None, None, None
        pmc_files = PMC_SUMMARIES if ((S, A, B) == (17, 6, 256) and args.precision == "f32") else (
            PMC_SUMMARIES_C5 if ((S, A, B) == (39, 28, 1024) and lb) else [])
        if pmc_files:
            for path in pmc_files:
                if not os.path.exists(path):
                    continue
                with open(path) as fh:
                    for name, rec in json.load(fh).items():
                        if kernel_name in name:
                            traffic = rec.get("hbm_bytes_per_launch_corrected")
                            mfma_util = rec.get("mfma_util")      # SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz)
                            pmc_src = os.path.relpath(path, ROOT)
                if pmc_src:
                    break
        # f32: the fp32-MFMA peak.  bf16 mode: the share of the launch's FLOPs that runs on v_mfma_f32_*_bf16 (layer 0 /
        # layer 1 forward, dW1, dH0, dW0) is priced against the bf16 peak, the rest (heads, the policy's dH1) against fp32
        peak = PEAK_F32_MFMA_TFLOPS
        peak_step = peak
        if args.precision == "bf16" and not lb:
            fp32_share = (2 * B * HID * (3 + 2 * A) * 2) / f_bwd            # dY.W2 and dW2 of the four heads
            peak = peak_step = 1.0 / (fp32_share / PEAK_F32_MFMA_TFLOPS + (1.0 - fp32_share) / PEAK_BF16_MFMA_TFLOPS)
        elif lb:
            # large-batch path: every product of the three kernels runs on v_mfma_f32_16x16x32_bf16 except the scalar
            # nets' heads and their dY . W2 / dW2 (vector ALU, ~0.1 % of the FLOPs): priced against the bf16 peak
            peak = peak_step = PEAK_BF16_MFMA_TFLOPS
        roof = {"bound": "mfma", "kernel": kernel_name, "achieved": round(ach, 3), "peak": round(peak, 1),
                "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                "traffic": None if traffic is None else round(traffic),
                "traffic_kind": "fabric bytes per launch of this kernel (L2 <-> memory side, Infinity-Cache hits included), not HBM bytes; "
                                f"compulsory HBM bytes per step = the row gather = batch x row stride = {B * ((2 * S + A + 2 + 3) // 4 * 4) * 4} B "
                                "(parameters, optimiser state and activations are cache-resident)",
                "traffic_source": None if traffic is None else f"{pmc_src} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, 2*FETCH+WRITE; not measured in this run)",
                "mfma_util_pmc": None if mfma_util is None else round(mfma_util, 4),
                "mfma_util_source": None if mfma_util is None else pmc_src,
                "flops_per_launch": f_dom, "avg_launch_us": round(t_dom, 3),
                "avg_launch_us_source": "HIP events around 500 back-to-back launches of the kernel, in this run",
                "kernel_us": kernel_us,
                "step_flops": f_fwd + f_bwd,
                # the same fraction for the whole step, from the TIMED region (all launches, gaps and fixed costs included)
                "frac_step": round((f_fwd + f_bwd) * args.steps / dt / 1e12 / peak_step, 4)}

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return 0

    value = args.steps * world / dt
    cfg = {"workload": f"IQL step on synthetic buffer (obs={S}, act={A}, {rows} rows), batch={B} per GPU",
           "global_batch": B * world, "parallelism": f"dp{world}" if world > 1 else "single",
           "global_steps_per_s": round(args.steps / dt, 1)}
    # prepare_train_steps (before the --warmup steps, outside the timed region) REHEARSES every chunk graph once (64 + 16 + 4
    # + 2 + 1 steps, parameters saved and restored) and then replays the 64-step chunk IQLHIP_PREPARE_WARM_CHUNKS times
    # (default 16) to ramp the clocks: the timed region starts behind that many untimed steps plus --warmup
    if not eager_dp[0]:
        warm_chunks = 16 if world == 1 else 0
        if os.environ.get("IQLHIP_PREPARE_WARM_CHUNKS") is not None:
            warm_chunks = max(0, int(os.environ["IQLHIP_PREPARE_WARM_CHUNKS"])) if world == 1 else 0
        cfg["prepare_rehearsal_steps"] = 64 + 16 + 4 + 2 + 1 + 64 * warm_chunks
    cfg["startup_phases_s"] = phases       # untimed start-up work, wall-clock on rank 0 (import and process launch not included)
    if repeats > 1:
        cfg["timed_regions_ms"] = [round(x * 1e3, 3) for x in dts]
        cfg["reported"] = f"median of {repeats} timed regions of {args.steps} steps"
    if world > 1:
        cfg["exchange"] = exchange
        cfg["exchange_why"] = why
        if probe:
            cfg["exchange_probe"] = probe
        refused = {m: v["unavailable"] for m, v in probe.items() if isinstance(v, dict) and "unavailable" in v}
        cfg["multi_gpu_note"] = (("ranks on distinct GPUs" if torch.cuda.device_count() >= world
                                  else f"REHEARSAL: {world} ranks share {torch.cuda.device_count()} GPU(s) — not a scaling measurement")
                                 + f"; exchanges attached: {attached if want == 'auto' else [want]}; refused: {refused if refused else 'none'}"
                                 + f"; timed region ran on: {exchange}")
    out = {
        "metric": f"IQL gradient-steps/sec at batch={B} (D4RL obs/act dims)",      # BASELINE.json's metric at the default B = 256
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
        "config": cfg,
    }
    if roof:
        out["roofline"] = roof
    if world == 1 and not args.no_cpu_baseline:
        from oracle import iql_torch_port as port
        try:
            ncpu = len(os.sched_getaffinity(0))
        except AttributeError:
            ncpu = os.cpu_count() or 1
        cpu_rows = min(rows, 1_000_000)
        # threads: 1 and the box's CPU share for one GPU (16; os.sched_getaffinity reports every core of the host, but a
        # 1-GPU box is limited to 16 CPUs' worth of time — oversubscribing that with hundreds of OpenMP threads stalls)
        thr_many = min(ncpu, 16)
        runs = []
        for thr in sorted({1, thr_many}):
            sys.stderr.write(f"bench.py: cpu_baseline on {thr} thread(s), {args.cpu_seconds:.0f} s ...\n")
            sys.stderr.flush()
            sps, n, el = port.time_cpu_steps(S, A, B, cpu_rows, seconds_budget=args.cpu_seconds, threads=thr,
                                             warmup=200, max_steps=2000)
            runs.append((sps, thr, n, el))
        best, cores, _, _ = max(runs)
        out["cpu_baseline"] = {
            "value": round(best, 2), "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"PyTorch-CPU port of the reference step (oracle/iql_torch_port.py), same S/A/B, {cpu_rows}-row buffer, 200 warm-up steps then <= 2000 steps or {args.cpu_seconds:.0f} s per thread setting: "
                      + "; ".join(f"{n} steps in {el:.1f}s @{thr} thread(s) = {sps:.1f}/s" for sps, thr, n, el in runs)
                      + f"; torch {torch.__version__}; the process may run on {ncpu} host cpus, of which a 1-GPU box grants 16"}
        out["speedup_vs_cpu"] = round(value / best, 1)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
