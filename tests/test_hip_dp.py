"""GPU tests of the multi-step driver's chunking and of the in-library gradient exchanges (RCCL all-reduce and the
direct peer-read exchange), all through the C ABI.  Multi-rank cases start fresh child processes (before this
process's children touch the GPU themselves) that share the one GPU of the test box."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import synth
from helpers import ROOT, load_golden, sub

pytestmark = pytest.mark.gpu

HYPER = {"iql_tau": 0.7, "beta": 3.0, "discount": 0.99, "tau": 0.005}
LRS = {"v": 3e-4, "q": 3e-4, "pi": 3e-4}


def _mk(seed, S=17, A=6, N=5000, max_steps=1000):
    import iql
    from hip_helpers import build_hip_trainer
    params = synth.synth_params(S, A, seed=seed)
    data = synth.synth_transitions(N, S, A, seed=seed + 1)
    buf = iql.ReplayBuffer(S, A, N, "cuda")
    buf.load_d4rl_dataset({k: v.copy() for k, v in data.items()})
    return params, buf, (lambda: build_hip_trainer(params, S, A, True, HYPER, LRS, max_steps))


def _same_params(a, b):
    from hip_helpers import read_params
    pa, pb = read_params(a), read_params(b)
    for n in pa:
        for k in pa[n]:
            assert np.array_equal(pa[n][k], pb[n][k]), (n, k)


@pytest.mark.parametrize("K,B", [(64, 256), (70, 256), (100, 256), (129, 33), (200, 100)])
def test_train_steps_chunks_equal_eager_steps(K, B):
    """A run that spans graph chunks (64 steps each) and directly launched remainder steps equals K eager steps on
    the same device index stream, bitwise: losses of every step, parameters, LR schedule."""
    import iqlhip_binding as hb
    params, buf, new = _mk(101)
    g = new()
    losses = g.train_steps(buf, K, B, seed=77)
    assert losses.shape == (K, 3) and np.all(np.isfinite(losses)) and g.total_it == K
    e = new()
    idx = torch.empty(K * B, dtype=torch.int64, device="cuda")
    hb.check(hb.lib().iqlhip_draw_indices(idx.data_ptr(), K * B, buf._size, 77, 0, torch.cuda.current_stream().cuda_stream))
    for k in range(K):
        log = e.train(buf.gather(idx[k * B:(k + 1) * B]))
        assert [log["value_loss"], log["q_loss"], log["actor_loss"]] == [float(x) for x in losses[k]], k
    _same_params(g, e)
    assert g.actor_optimizer.param_groups[0]["lr"] == e.actor_optimizer.param_groups[0]["lr"]


def test_train_steps_split_calls_equal_one_call():
    """The result does not depend on how a run is cut into calls (20 + 5 + 64 + 41 == 130): the index stream position,
    the scalar tables (incl. the look-ahead cache) and the loss ring line up across calls."""
    params, buf, new = _mk(111)
    a, b = new(), new()
    la = a.train_steps(buf, 130, 256, seed=3)
    parts = [b.train_steps(buf, n, 256, seed=3) for n in (20, 5, 64, 41)]
    assert np.array_equal(la, np.concatenate(parts))
    _same_params(a, b)


def test_prepare_then_short_runs_never_capture():
    """prepare_train_steps builds the chunk graph up front; runs shorter than a chunk launch directly."""
    params, buf, new = _mk(121)
    t = new()
    t.prepare_train_steps(buf, 256)
    l1 = t.train_steps(buf, 20, 256, seed=1)
    u = new()
    l2 = u.train_steps(buf, 20, 256, seed=1)
    assert np.array_equal(l1, l2)


@pytest.fixture
def gloo_world1():
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(29600 + os.getpid() % 1000)
    dist.init_process_group("gloo", rank=0, world_size=1)
    yield
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["rccl", "p2p"])
def test_exchange_world1_equals_plain_steps(gloo_world1, exchange):
    """With a 1-rank group the exchange path (flatten -> ncclAllReduce captured in the chunk graph / flag handshake +
    peer-read update) lands on the same parameters and losses as the plain three-launch step, bitwise, for eager
    train() and for train_steps across a chunk boundary."""
    params, buf, new = _mk(131)
    plain, dpt = new(), new()
    dpt.enable_data_parallel(exchange=exchange)
    dpt.prepare_train_steps(buf, 256)        # rehearses the captured chunks (exchange included) without training
    rehearsed = dpt.exchange_status()["steps"]
    once = 64 + 16 + 4 + 2 + 1                  # every chunk size once (a call's head steps are launched directly: nothing to rehearse)
    assert rehearsed == {"rccl": once, "p2p": 2 * once}[exchange]           # (P2P: once per buffer parity)
    batch = buf.gather(torch.arange(256, device="cuda"))
    assert plain.train(batch) == dpt.train(batch)
    la = plain.train_steps(buf, 70, 256, seed=5)
    lb = dpt.train_steps(buf, 70, 256, seed=5)
    assert np.array_equal(la, lb)
    _same_params(plain, dpt)
    st = dpt.exchange_status()
    assert st["timed_out_step"] == 0 and st["steps"] == rehearsed + 71
    hb_mode = {"rccl": 1, "p2p": 2}[exchange]
    assert st["mode"] == hb_mode


def _run_ranks(scenario, world, tmp_path, timeout=300):
    port = str(29700 + os.getpid() % 1000)
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    worker = os.path.join(ROOT, "tests", "dp_rank_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, scenario, str(r), str(world), port, str(tmp_path)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=timeout)       # the host-side bound: a stuck exchange fails the test
            outs.append(o.decode("utf-8", "replace"))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} exited {p.returncode}:\n{outs[r][-3000:]}"
    return [json.load(open(os.path.join(tmp_path, f"rank{r}.json"))) for r in range(world)]


def test_p2p_two_ranks_reproduce_big_batch_fixture(tmp_path):
    """Two processes on the one GPU, exchange blocks mapped through hipIpc: one DP step on the two halves of the
    reference's B=2048 batch equals the reference's single-process step (fixture g8); after 70 more steps through
    train_steps (graph chunk + direct steps, flag handshake inside) the replicas are still bit-identical."""
    res = _run_ranks("fixture", 2, tmp_path)
    z, meta = load_golden("g8_dp_B2048")
    for r in res:
        np.testing.assert_allclose(r["losses"], z["losses"], rtol=1e-5)
        assert r["status"]["timed_out_step"] == 0 and r["status"]["steps"] == 71 and r["free_losses_finite"]
    assert res[0]["losses"] == res[1]["losses"] and res[0]["free_losses_last"] == res[1]["free_losses_last"]
    for tag in ("step1", "step71"):
        a = np.load(os.path.join(tmp_path, f"{tag}_rank0.npz"))
        b = np.load(os.path.join(tmp_path, f"{tag}_rank1.npz"))
        for k in a.files:
            assert np.array_equal(a[k], b[k]), (tag, k)
    a = np.load(os.path.join(tmp_path, "step1_rank0.npz"))
    for net in ("vf", "q1", "q2", "pi", "qt1", "qt2"):
        for t in ("w0", "w1", "b1", "w2"):
            key = f"param.{net}.{t}"
            if key not in z:
                continue
            want = z[key]
            got = sub(a[f"{net}.{t}"], meta["stride"]).reshape(want.shape)
            tol = 1e-6 if net.startswith("qt") else 2e-6
            assert np.max(np.abs(got.astype(np.float64) - want)) <= tol, key


def test_p2p_absent_peer_times_out_instead_of_hanging(tmp_path):
    """A peer that never signals: the in-stream wait gives up after its timeout (300 ms here), the stream drains, the
    status word names the step — nothing hangs and the process exits cleanly."""
    res = _run_ranks("absent", 2, tmp_path, timeout=120)
    assert res[0]["status"]["timed_out_step"] == 1 and res[0]["status"]["steps"] == 3


def _bench_two_ranks_one_gpu(extra_env, args=("--rows", "200000", "--steps", "40", "--warmup", "32"), timeout=420):
    env = dict(os.environ)
    env.update({"HSA_ENABLE_IPC_MODE_LEGACY": "0", "IQLHIP_DIST_BACKEND": "gloo", "IQLHIP_PREPARE_WARM_CHUNKS": "0"})
    env.update(extra_env)
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline",
           "--master-port", str(29300 + os.getpid() % 500)] + list(args)
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    assert out.returncode == 0, out.stderr.decode("utf-8", "replace")[-3000:]
    line = [l for l in out.stdout.decode().splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_bench_recovers_from_an_exchange_probe_that_left_the_replicas_diverged():
    """`bench.py --gpus 2 --exchange auto` when the peer-to-peer probe leaves the replicas REALLY diverged (test hook:
    the last rank perturbs its parameters after the probe run) and RCCL is unavailable (two ranks on this one GPU): the
    run must re-synchronise rank 0's arenas / step counters, clear the exchange status, fall back to the eager
    torch.distributed exchange and finish with identical replicas (bench.py asserts that itself before it prints)."""
    d = _bench_two_ranks_one_gpu({"IQLHIP_BENCH_BREAK_PROBE": "p2p"})
    cfg = d["config"]
    assert cfg["exchange"] == "torch" and "no in-library exchange" in cfg["exchange_why"]
    assert cfg["exchange_probe"]["p2p"]["replicas_equal"] is False and cfg["exchange_probe"]["p2p"]["resynced"] is True
    assert "unavailable" in cfg["exchange_probe"]["rccl"]
    assert d["n_gpus"] == 2 and d["value"] > 0 and "REHEARSAL" in cfg["multi_gpu_note"]


def test_bench_two_ranks_on_one_gpu_pick_the_peer_exchange_and_say_why():
    d = _bench_two_ranks_one_gpu({})
    cfg = d["config"]
    assert cfg["exchange"] == "p2p" and cfg["exchange_probe"]["p2p"]["replicas_equal"] is True
    assert "only in-library exchange" in cfg["exchange_why"]


def test_head_modes_and_forward_layouts_agree(tmp_path):
    """How a train_steps call starts (IQLHIP_HEAD = direct / graph / plain, IQLHIP_DIRECT_ALL) and how many slices a forward
    or backward block walks are placement and launch choices only: losses of every step and the parameters after 98
    steps are identical, bit for bit, across all of them (each variant in a fresh process: the switches are read when the
    library is loaded)."""
    variants = [{}, {"IQLHIP_HEAD": "graph"}, {"IQLHIP_HEAD": "plain"}, {"IQLHIP_DIRECT_ALL": "1"},
                {"IQLHIP_FWD_SPB_L2": "2", "IQLHIP_BWD_SPB_L2": "2"}]
    worker = os.path.join(ROOT, "tests", "head_mode_worker.py")
    outs = []
    for i, extra in enumerate(variants):
        env = dict(os.environ)
        env.update(extra)
        path = os.path.join(tmp_path, f"v{i}.json")
        r = subprocess.run([sys.executable, worker, path], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        assert r.returncode == 0, r.stdout.decode("utf-8", "replace")[-3000:]
        outs.append(json.load(open(path)))
    for i, o in enumerate(outs[1:], 1):
        assert o["losses"] == outs[0]["losses"], variants[i]
        assert o["params"] == outs[0]["params"], variants[i]
    assert len(outs[0]["losses"]) == 98 and np.all(np.isfinite(np.array(outs[0]["losses"])))


@pytest.mark.parametrize("exchange", ["rccl", "p2p"])
def test_exchange_world1_equals_plain_steps_on_the_large_batch_bf16_path(gloo_world1, exchange):
    """BASELINE configs[4] is the data-parallel bf16 run: 1 024 rows per rank take the large-batch kernels
    (csrc/iqlhip_lb_kernels.h), whose gradient reaches the update kernel flat (flatten -> exchange) instead of through the
    chunk-group slabs, and whose update must still keep the forward's operand images current.  With a 1-rank group both
    exchanges land on the plain step's losses and parameters bit for bit, eagerly and through the chunk graphs."""
    S, A, B = 39, 28, 1024
    params, buf, _ = _mk(141, S=S, A=A, N=4000)
    from hip_helpers import build_hip_trainer
    hyper = {"iql_tau": 0.8, "beta": 3.0, "discount": 0.99, "tau": 0.005}
    new = lambda: build_hip_trainer(params, S, A, True, hyper, LRS, 1000)
    plain, dpt = new(), new()
    plain.set_precision("bf16")
    dpt.set_precision("bf16")
    dpt.enable_data_parallel(exchange=exchange)
    batch = buf.gather(torch.arange(B, device="cuda"))
    for _ in range(3):                       # (three steps: a stale operand image would show from the second on)
        assert plain.train(batch) == dpt.train(batch)
    la = plain.train_steps(buf, 6, B, seed=5)
    lb = dpt.train_steps(buf, 6, B, seed=5)
    assert np.array_equal(la, lb)
    _same_params(plain, dpt)
    assert dpt.exchange_status()["timed_out_step"] == 0
