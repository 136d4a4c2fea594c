"""One rank of a multi-process data-parallel test (started as a fresh child process by tests/test_hip_dp.py — before
anything in it has touched the GPU).  Several ranks may share ONE GPU: the peer-to-peer exchange only needs the ranks'
exchange blocks mapped into each other through hipIpc, which works within one device too; the rendezvous and the
one-off state broadcast go over gloo (CPU), so no collective library needs two ranks on one device.

    python dp_rank_worker.py <scenario> <rank> <world> <port> <out_dir>

Scenarios
  fixture   one DP step on this rank's slice of the reference's B=2048 fixture batch (g8), then 70 more steps on a
            buffer through train_steps (one graph chunk + 6 direct steps, exchange inside); dumps losses / params.
  absent    rank 1 never steps: rank 0's in-stream wait must give up after its timeout and report it.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jsrl-corl_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    scenario, rank, world, port, out_dir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    import numpy as np
    import torch
    import torch.distributed as dist
    import datetime

    import iql
    import iqlhip_dp as dp
    import synth
    from helpers import load_golden, single_step_inputs
    from hip_helpers import build_hip_trainer, read_params, to_torch_batch

    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    torch.cuda.set_device(0)
    z, meta = load_golden("g8_dp_B2048")
    params, batch, hyper = single_step_inputs(meta)
    S, A, B = meta["S"], meta["A"], meta["B"]
    tr = build_hip_trainer(params, S, A, meta["gaussian"], hyper, meta["lrs"], meta["max_steps"], device="cuda:0")
    b = B // world
    tr.reserve_batch(b)
    out = {"rank": rank}
    if scenario == "fixture":
        tr.enable_data_parallel(exchange="p2p", timeout_ms=20000)
        sl = dp.shard(B, rank, world)
        local = {k: v[sl] for k, v in batch.items()}
        log = tr.train(to_torch_batch(local, "cuda:0"))
        out["losses"] = [log["value_loss"], log["q_loss"], log["actor_loss"]]
        p1 = read_params(tr)
        np.savez(os.path.join(out_dir, f"step1_rank{rank}.npz"),
                 **{f"{n}.{k}": v for n, t in p1.items() for k, v in t.items()})
        # 70 more steps from a (replicated) buffer: every rank draws its own rows, one 64-step graph chunk + 6 direct
        N = 4096
        data = synth.synth_transitions(N, S, A, seed=77)
        buf = iql.ReplayBuffer(S, A, N, "cuda:0")
        buf.load_d4rl_dataset(data)
        losses = tr.train_steps(buf, 70, 256, seed=9)
        out["free_losses_finite"] = bool(np.all(np.isfinite(losses)))
        out["free_losses_last"] = [float(x) for x in losses[-1]]
        st = tr.exchange_status()
        out["status"] = st
        p2 = read_params(tr)
        np.savez(os.path.join(out_dir, f"step71_rank{rank}.npz"),
                 **{f"{n}.{k}": v for n, t in p2.items() for k, v in t.items()})
        dist.barrier()
    elif scenario == "absent":
        tr.enable_data_parallel(exchange="p2p", timeout_ms=300)
        if rank == 0:
            sl = dp.shard(B, rank, world)
            local = {k: v[sl] for k, v in batch.items()}
            tb = to_torch_batch(local, "cuda:0")
            for _ in range(3):                       # the first wait times out; the later ones return at once (sticky)
                tr._prepare(b)
                bs, keep, n = tr._batch_struct(tb)
                tr._run_step(bs, n, sync=False)
            st = tr.exchange_status()
            out["status"] = st
        dist.barrier()
    else:
        raise SystemExit(f"unknown scenario {scenario}")
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(out, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
