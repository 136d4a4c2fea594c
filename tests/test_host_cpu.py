"""CPU-only tests (no GPU in the process): the C-ABI library loads and exports every
symbol include/iqlhip.h declares, the arena layout is what the shim relies on, and the
host logic of the drop-in classes (ReplayBuffer bookkeeping, checkpoint format, LR
schedule fast path, config dataclasses, error behaviour) matches the reference's,
pinned by the reference-generated fixtures g3..g6."""
import json
import os
import re

import numpy as np
import pytest
import torch

import iql
import iqlhip_binding as hb
import synth
from helpers import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------- C ABI
def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "iqlhip.h")).read()
    declared = set(re.findall(r"\b(iqlhip_[a-z_0-9]+)\s*\(", header))
    declared -= {"iqlhip_ctx"}
    bound = {name for name, _, _ in hb.SYMBOLS}
    assert declared == bound, f"header vs binding mismatch: {declared ^ bound}"
    lib = hb.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.iqlhip_version() >= 100


def test_arena_layout_is_consistent():
    for (S, A, gauss) in ((17, 6, True), (29, 8, False), (39, 28, True), (3, 2, True)):
        L = hb.arena_layout(S, A, gauss)
        prev_end = 0
        for i, nl in enumerate(L.net):
            k = S + A if i in (hb.NET_Q1, hb.NET_Q2) else S
            d = A if i == hb.NET_PI else 1
            assert (nl.k_in, nl.d_out) == (k, d)
            assert nl.seg_begin == prev_end and nl.seg_begin % 64 == 0 and nl.seg_end % 64 == 0
            assert nl.w1 == nl.seg_begin and nl.w0 == nl.w1 + 256 * 256 and nl.b0 == nl.w0 + 256 * k
            assert nl.b1 == nl.b0 + 256 and nl.w2 == nl.b1 + 256 and nl.b2 == nl.w2 + d * 256
            for off in (nl.w0, nl.b0, nl.w1, nl.b1, nl.w2, nl.b2):
                assert off % 4 == 0
            if i == hb.NET_PI and gauss:
                assert nl.log_std == nl.b2 + (d + 3) // 4 * 4
            else:
                assert nl.log_std == -1
            prev_end = nl.seg_end
        assert L.n_params == prev_end
        assert L.target_src == L.net[hb.NET_Q1].seg_begin
        assert L.n_target == L.net[hb.NET_Q2].seg_end - L.net[hb.NET_Q1].seg_begin
    assert hb.row_stride(17, 6) == 44 and hb.row_stride(29, 8) == 68 and hb.row_stride(39, 28) == 108


def test_error_codes_map_to_reference_exception_types():
    with pytest.raises(NotImplementedError):
        hb.arena_layout(17, 6, True, hidden_dim=128)
    with pytest.raises(NotImplementedError):
        hb.arena_layout(100, 40, True)
    with pytest.raises(ValueError):
        hb.arena_layout(0, 6, True)


def test_argument_validation_precedes_any_device_work():
    """Entry points reject bad arguments before touching the GPU (so these run on a CPU-only host)."""
    lib = hb.lib()
    # NULL pointers / bad geometry -> IQLHIP_EINVAL -> ValueError
    for call in (
        lambda: lib.iqlhip_rows_gather_packed(None, 44, 100, None, 4, None, None),
        lambda: lib.iqlhip_rows_gather_packed(16, 43, 100, 16, 4, 16, None),      # stride not a multiple of 4 floats
        lambda: lib.iqlhip_rows_gather_packed(8, 44, 100, 16, 4, 16, None),       # rows not 16-byte aligned
        lambda: lib.iqlhip_rows_gather_packed(16, 44, 0, 16, 4, 16, None),        # a buffer of no rows
        lambda: lib.iqlhip_rows_gather_packed_h(16, 44, 100, None, None, 4, 16, None),
        lambda: lib.iqlhip_actor_forward(None, None, 17, 1, None, 6, 1.0, None, 6, None),
        lambda: lib.iqlhip_rows_gather(None, 44, 100, 17, 6, None, 4, None, None, None, None, None, None),
        lambda: lib.iqlhip_draw_indices(None, 4, 10, 0, 0, None),
    ):
        with pytest.raises(ValueError):
            hb.check(call())
    assert "argument" in hb.last_error().lower() or hb.last_error()
    # zero rows are a no-op, not an error
    hb.check(lib.iqlhip_rows_gather_packed(16, 44, 100, 16, 0, 16, None))
    # a host-visible row index outside the buffer -> IQLHIP_EINDEX -> IndexError (the reference's tensor indexing,
    # iql.py:173-177), before any launch: the entry points that receive the indices in host memory
    import ctypes as C
    bad = (C.c_int64 * 4)(0, 5, 100, 7)
    neg = (C.c_int64 * 2)(3, -1)
    for call in (
        lambda: lib.iqlhip_rows_sample_packed(16, 44, 100, C.addressof(bad), 4, 16, None),
        lambda: lib.iqlhip_rows_gather_packed_h(16, 44, 100, C.addressof(neg), 16, 2, 16, None),
    ):
        with pytest.raises(IndexError):
            hb.check(call())
    assert "out of bounds" in hb.last_error()


# ---------------------------------------------------------------- replay buffer host logic
def test_replay_buffer_sample_matches_reference_fixture():
    z, meta = load_golden("g3_gather")
    data = synth.synth_transitions(meta["N"], meta["S"], meta["A"], seed=meta["data_seed"])
    buf = iql.ReplayBuffer(meta["S"], meta["A"], meta["capacity"], "cpu")
    buf.load_d4rl_dataset({k: v.copy() for k, v in data.items()})
    assert (buf._size, buf._pointer) == (meta["size"], meta["pointer"])
    np.random.seed(meta["np_seed"])
    s, a, r, ns, d = buf.sample(meta["B"])
    for got, key in ((s, "s"), (a, "a"), (r, "r"), (ns, "ns"), (d, "d")):
        assert tuple(got.shape) == z[key].shape
        assert np.array_equal(got.numpy(), z[key]), key
    # the reference's attribute views
    assert buf._states.shape == (meta["capacity"], meta["S"]) and buf._rewards.shape == (meta["capacity"], 1)
    assert np.array_equal(buf._states[: meta["N"]].numpy(), data["observations"])
    assert np.array_equal(buf._dones[: meta["N"], 0].numpy(), data["terminals"])


def test_replay_buffer_ring_insert_matches_reference_fixture():
    z, meta = load_golden("g4_ring")
    S, A, cap = meta["S"], meta["A"], meta["capacity"]
    data = synth.synth_transitions(5, S, A, seed=meta["load_seed"])
    extra = synth.synth_transitions(7, S, A, seed=meta["extra_seed"])
    buf = iql.ReplayBuffer(S, A, cap, "cpu")
    buf.load_d4rl_dataset({k: v.copy() for k, v in data.items()})
    trace = [(buf._pointer, buf._size)]
    for i in range(7):
        buf.add_transition(extra["observations"][i], extra["actions"][i], float(extra["rewards"][i]),
                           extra["next_observations"][i], bool(extra["terminals"][i] > 0.5 or i == 2))
        trace.append((buf._pointer, buf._size))
    assert np.array_equal(np.array(trace), z["trace"])
    for attr, key in (("_states", "states"), ("_actions", "actions"), ("_rewards", "rewards"),
                      ("_next_states", "next_states"), ("_dones", "dones")):
        assert np.array_equal(getattr(buf, attr).numpy(), z[key]), key
    with pytest.raises(ValueError) as e1:
        buf.load_d4rl_dataset(data)
    assert str(e1.value) == meta["errors"]["nonempty"]
    with pytest.raises(ValueError) as e2:
        iql.ReplayBuffer(S, A, 3, "cpu").load_d4rl_dataset(data)
    assert str(e2.value) == meta["errors"]["too_small"]


def test_offline_buffer_flavour():
    buf = iql.OfflineReplayBuffer(3, 2, 16, "cpu")
    buf.load_d4rl_dataset(synth.synth_transitions(10, 3, 2, seed=1))
    assert buf._index_bound() == 10
    with pytest.raises(NotImplementedError):
        buf.add_transition()


# ---------------------------------------------------------------- trainer host logic
def _cpu_trainer(S=17, A=6, gaussian=True, max_steps=1000):
    qf, vf = iql.TwinQ(S, A), iql.ValueFunction(S)
    actor = (iql.GaussianPolicy if gaussian else iql.DeterministicPolicy)(S, A, 1.0)
    return iql.ImplicitQLearning(
        1.0, actor, torch.optim.Adam(actor.parameters(), lr=3e-4), qf, torch.optim.Adam(qf.parameters(), lr=3e-4),
        vf, torch.optim.Adam(vf.parameters(), lr=3e-4), max_steps=max_steps, device="cpu")


def test_state_dict_format_matches_reference_fixture():
    z, meta = load_golden("g6_statedict")
    tr = _cpu_trainer()
    sd = tr.state_dict()
    assert list(sd.keys()) == meta["top_keys"]
    for k in ("qf", "vf", "actor"):
        assert {kk: list(vv.shape) for kk, vv in sd[k].items()} == meta[k]
    assert sorted(sd["actor_lr_schedule"].keys()) == sorted(meta["actor_lr_schedule"].keys())
    for k in ("q_optimizer", "v_optimizer", "actor_optimizer"):
        assert sorted(sd[k]["param_groups"][0].keys()) == meta[k]["param_group_keys"]
        assert sd[k]["param_groups"][0]["params"] == meta[k]["params"]
    # round trip + the reference's quirk: after load the target equals qf and is trainable-flagged
    tr2 = _cpu_trainer()
    tr2.load_state_dict(sd)
    assert all(torch.equal(a, b) for a, b in zip(tr2.q_target.parameters(), tr2.qf.parameters()))
    assert next(tr2.q_target.parameters()).requires_grad == meta["target_requires_grad_after_load"]
    tr3 = _cpu_trainer()
    before = [p.clone() for p in tr3.q_optimizer.param_groups[0]["params"]]
    tr3.partial_load_state_dict(sd)
    assert tr3.total_it == sd["total_it"]
    assert all(torch.equal(a, b) for a, b in zip(tr3.qf.parameters(), tr.qf.parameters()))
    assert len(tr3.q_optimizer.state) == 0 and len(before) == 12


def test_cpu_device_has_no_step():
    tr = _cpu_trainer()
    with pytest.raises(RuntimeError, match="GPU"):
        tr.train([torch.zeros(4, 17), torch.zeros(4, 6), torch.zeros(4, 1), torch.zeros(4, 17), torch.zeros(4, 1)])


def test_fast_lr_schedule_equals_torch_scheduler():
    z, meta = load_golden("g5_lr")
    tr = _cpu_trainer(3, 2, True, max_steps=meta["T"])
    got = np.concatenate([tr._advance_schedule(700), tr._advance_schedule(801)])
    assert np.array_equal(got, z["lrs"])          # bit-equal to the reference's recursion, incl. the restart branch
    assert tr.actor_lr_schedule.last_epoch == 1501
    ref = _cpu_trainer(3, 2, True, max_steps=meta["T"])
    for _ in range(1501):
        ref._step_schedule()
    assert ref.actor_lr_schedule.state_dict() == tr.actor_lr_schedule.state_dict()
    none = _cpu_trainer(3, 2, True, max_steps=None)
    assert none.actor_lr_schedule is None and np.all(none._advance_schedule(5) == 3e-4)


def test_scalar_table_matches_torch_adam_scalars():
    from oracle import iql_oracle as O
    tr = _cpu_trainer(3, 2, True, max_steps=None)
    tab = tr._scalar_table(10, 1.0 / 256)
    for t in range(1, 11):
        step_size, bc2 = O.adam_scalars(3e-4, t)
        assert tab[t - 1, 0] == np.float32(step_size) and tab[t - 1, 3] == np.float32(bc2)
    assert tab[0, 6] == np.float32(0.999) and tab[0, 7] == np.float32(1 - 0.9) and tab[0, 11] == np.float32(1 / 256)
    assert tr._adam_t == {"v": 10, "q": 10, "pi": 10}


def test_lookahead_scalar_table_equals_direct_computation():
    """The table rows computed ahead of time (while the GPU runs the previous chunk) and sliced to the next call's
    length are bit-equal to rows computed on demand, for any sequence of call lengths, across the cosine restart."""
    a = _cpu_trainer(3, 2, True, max_steps=100)
    b = _cpu_trainer(3, 2, True, max_steps=100)
    for k in (5, 20, 64, 1, 70, 130, 3):
        ta = a._scalar_table(k, 1.0 / 256)
        a._lookahead_table(k, 1.0 / 256)
        tb = b._scalar_table(k, 1.0 / 256)               # never looks ahead
        assert ta.shape[0] >= k and tb.shape[0] == k      # (a look-ahead table is handed over whole: its first k rows count)
        assert b._table_cache is None and np.array_equal(ta[:k], tb), k
        assert a.actor_lr_schedule.state_dict() == b.actor_lr_schedule.state_dict()
        assert a._adam_t == b._adam_t
    # a changed learning rate invalidates the look-ahead
    a._lookahead_table(10, 1.0 / 256)
    for t in (a, b):
        t.v_optimizer.param_groups[0]["lr"] = 1e-3
    assert np.array_equal(a._scalar_table(10, 1.0 / 256)[:10], b._scalar_table(10, 1.0 / 256))


def test_bench_launcher_fans_out_one_rank_per_gpu_without_touching_the_gpu():
    """`python bench.py --gpus N` with no launcher in the environment starts N fresh rank processes itself (dry run:
    the plan as data); a launcher world size that contradicts --gpus is refused; configs[3] rows at N > 1."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    bench = os.path.join(ROOT, "bench.py")
    out = subprocess.run([sys.executable, bench, "--gpus", "8", "--steps", "20", "--warmup", "5", "--dry-run-launch"],
                         env=env, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    plan = json.loads(out.stdout)
    assert plan["self_launch"] and plan["gpus"] == 8 and plan["rows"] == 10_000_000 and len(plan["ranks"]) == 8
    ports = {r["env"]["MASTER_PORT"] for r in plan["ranks"]}
    assert len(ports) == 1
    for i, r in enumerate(plan["ranks"]):
        assert r["rank"] == i and r["env"]["RANK"] == str(i) and r["env"]["LOCAL_RANK"] == str(i)
        assert r["env"]["WORLD_SIZE"] == "8" and r["env"]["MASTER_ADDR"] == "127.0.0.1"
        assert r["cmd"][1] == bench and "--dry-run-launch" not in r["cmd"] and r["cmd"][2:6] == ["--gpus", "8", "--steps", "20"]
    one = json.loads(subprocess.run([sys.executable, bench, "--dry-run-launch"], env=env, capture_output=True, text=True,
                                    timeout=60).stdout)
    assert not one["self_launch"] and one["rows"] == 1_000_000 and one["ranks"] == []
    env2 = dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    bad = subprocess.run([sys.executable, bench, "--gpus", "8"], env=env2, capture_output=True, text=True, timeout=60)
    assert bad.returncode == 2 and "WORLD_SIZE=2" in bad.stderr


def test_offline_flavour_surface_matches_reference_fixture():
    """iql_offline.py (drop-in for algorithms/offline/iql.py) against fixture g15, generated from the reference's
    offline module: the `is not None` dropout gate (state_dict keys for dropout None / 0.0 / 0.1), the unconditional
    LR schedule in the checkpoint, TrainConfig's fields and defaults, the buffer's add_transition."""
    import dataclasses
    import iql_offline as off
    z, meta = load_golden("g15_offline_surface")
    for tag, d in (("none", None), ("zero", 0.0), ("p10", 0.1)):
        assert list(off.MLP([5, 7, 7, 3], dropout=d).state_dict().keys()) == meta["mlp_keys"][tag]
        assert list(off.GaussianPolicy(5, 3, 1.0, dropout=d).state_dict().keys()) == meta["policy_keys"][tag]["gauss"]
        assert list(off.DeterministicPolicy(5, 3, 1.0, dropout=d).state_dict().keys()) == meta["policy_keys"][tag]["det"]
    assert list(off.GaussianPolicy(5, 3, 1.0).state_dict().keys()) == meta["policy_default_dropout_keys"]
    # the finetune flavour differs exactly at dropout = 0.0 (gate `> 0.0`, finetune/iql.py:332)
    assert list(iql.MLP([5, 7, 7, 3], dropout=0.0).state_dict().keys()) == meta["mlp_keys"]["none"]
    assert isinstance(off.GaussianPolicy(5, 3, 1.0), iql.GaussianPolicy)          # isinstance checks keep working
    q, v, a = off.TwinQ(5, 3), off.ValueFunction(5), off.GaussianPolicy(5, 3, 1.0)
    tr = off.ImplicitQLearning(1.0, a, torch.optim.Adam(a.parameters(), lr=3e-4), q, torch.optim.Adam(q.parameters(), lr=3e-4),
                               v, torch.optim.Adam(v.parameters(), lr=3e-4), max_steps=10, device="cpu")
    sd = tr.state_dict()
    assert list(sd.keys()) == meta["state_dict_keys"]
    assert sorted(sd["actor_lr_schedule"].keys()) == meta["schedule_state_keys"]
    assert (getattr(tr, "partial_load_state_dict", None) is not None) == meta["has_partial_load"]
    tr2 = off.ImplicitQLearning(1.0, a, torch.optim.Adam(a.parameters(), lr=3e-4), q, torch.optim.Adam(q.parameters(), lr=3e-4),
                                v, torch.optim.Adam(v.parameters(), lr=3e-4), max_steps=10, device="cpu")
    tr2.load_state_dict(sd)
    cfg = off.TrainConfig()
    want = meta["train_config"]
    assert [f.name for f in dataclasses.fields(cfg)] == list(want.keys())
    for k, val in want.items():
        if k not in ("name", "checkpoints_path"):
            assert getattr(cfg, k) == val, k
    buf = off.ReplayBuffer(3, 2, 8, "cpu")
    with pytest.raises(NotImplementedError):
        buf.add_transition()
    assert meta["add_transition"] == "NotImplementedError"


def test_unsupported_configurations_fail_loudly():
    tr = _cpu_trainer()
    tr.v_optimizer.param_groups[0]["weight_decay"] = 0.1
    with pytest.raises(NotImplementedError):
        tr._adam_hyper()
    with pytest.raises(ValueError):
        iql.MLP([4])
    with pytest.raises(ValueError):
        iql.MLP([4, 8, 2], squeeze_output=True)
    # dropout layers appear exactly when the reference creates them (finetune flavour: > 0)
    assert not any(isinstance(m, torch.nn.Dropout) for m in iql.MLP([4, 8, 8, 2], dropout=0.0).modules())
    assert sum(isinstance(m, torch.nn.Dropout) for m in iql.MLP([4, 8, 8, 2], dropout=0.1).modules()) == 2


def test_train_config_fields_match_reference():
    cfg = iql.TrainConfig()
    want = ["device", "env", "seed", "eval_seed", "eval_freq", "n_episodes", "offline_iterations", "online_iterations",
            "checkpoints_path", "load_model", "actor_dropout", "buffer_size", "batch_size", "discount", "tau", "beta",
            "iql_tau", "expl_noise", "noise_clip", "iql_deterministic", "normalize", "normalize_reward", "vf_lr",
            "qf_lr", "actor_lr", "project", "group", "name"]
    assert list(cfg.__dataclass_fields__.keys()) == want
    assert cfg.name.startswith("IQL-antmaze-umaze-v2-") and len(cfg.name.split("-")[-1]) == 8
    assert (cfg.batch_size, cfg.tau, cfg.beta, cfg.iql_tau, cfg.buffer_size) == (256, 0.005, 3.0, 0.7, 2_000_000)
    off = iql.OfflineTrainConfig(checkpoints_path="/tmp/x")
    assert off.actor_dropout is None and off.max_timesteps == 1_000_000 and off.checkpoints_path.startswith("/tmp/x/IQL-")
    # kw_only subclassing as in jsrl_w_iql.py:46 keeps working
    from dataclasses import dataclass

    @dataclass(kw_only=True)
    class J(iql.TrainConfig):
        n_curriculum_stages: int = 10
    assert J(n_curriculum_stages=3).n_curriculum_stages == 3


def test_module_surface_has_every_name_the_jsrl_files_import():
    names = ["ENVS_WITH_GOAL", "DeterministicPolicy", "GaussianPolicy", "ImplicitQLearning", "ReplayBuffer",
             "TrainConfig", "TwinQ", "ValueFunction", "compute_mean_std", "is_goal_reached", "modify_reward",
             "modify_reward_online", "normalize_states", "set_env_seed", "set_seed", "wandb_init", "wrap_env", "Tuple",
             "nn", "MLP", "soft_update", "asymmetric_l2_loss", "eval_actor", "return_reward_range"]
    for n in names:
        assert hasattr(iql, n), n


def test_host_helpers():
    d = {"rewards": np.array([1.0, 2.0, 3.0, 4.0], dtype=np.float32), "terminals": np.array([0, 1, 0, 0])}
    assert iql.return_reward_range(d, 2) == (3.0, 7.0)
    ds = {"rewards": np.ones(4, dtype=np.float32), "terminals": np.zeros(4)}
    assert iql.modify_reward(ds, "antmaze-large-diverse-v2") == {} and np.all(ds["rewards"] == 0.0)
    assert iql.modify_reward_online(1.0, "antmaze-x") == 0.0
    m, s = iql.compute_mean_std(np.array([[0.0, 2.0], [2.0, 2.0]]), 1e-3)
    assert np.allclose(m, [1.0, 2.0]) and np.allclose(s, [1.001, 0.001])
    assert iql.is_goal_reached(0.0, {"success": True}) and not iql.is_goal_reached(0.0, {})


def test_return_reward_range_equals_the_sequential_definition():
    """iql.py:262-274 walks the dataset transition by transition; the build finds the episode ends per
    terminal-delimited run.  Same episodes, same returns, on random terminal / time-limit patterns — including the
    reference's ValueError when no episode ever ends."""
    def sequential(rew, term, limit):
        out, acc, steps = [], 0.0, 0
        for r, d in zip(rew.tolist(), term.tolist()):
            acc, steps = acc + r, steps + 1
            if d or steps == limit:
                out.append(acc)
                acc, steps = 0.0, 0
        return (min(out), max(out)) if out else None

    rng = np.random.default_rng(0)
    for _ in range(200):
        n, limit = int(rng.integers(1, 300)), int(rng.integers(1, 12))
        ds = {"rewards": rng.standard_normal(n).astype(np.float32),
              "terminals": rng.random(n) < rng.choice([0.0, 0.02, 0.3, 1.0])}
        want = sequential(ds["rewards"], ds["terminals"], limit)
        if want is None:
            with pytest.raises(ValueError):
                iql.return_reward_range(ds, limit)
        else:
            got = iql.return_reward_range(ds, limit)
            assert np.allclose(got, want, rtol=1e-12, atol=1e-12)


def test_bench_rank_supervisor_stops_everything_on_the_first_failure_and_on_timeout():
    """bench.py's parent of self-launched ranks polls ALL children: one rank failing early must not leave the others
    (and the parent) waiting in a collective until its timeout — they are terminated and the failure's code returned;
    a run that exceeds the overall bound is terminated with 124."""
    import subprocess
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    sleeper = [sys.executable, "-c", "import time; time.sleep(60)"]
    procs = [subprocess.Popen(sleeper), subprocess.Popen([sys.executable, "-c", "import sys, time; time.sleep(0.3); sys.exit(3)"]),
             subprocess.Popen(sleeper)]
    t0 = time.monotonic()
    assert bench.wait_ranks(procs, timeout_s=30.0) == 3
    assert time.monotonic() - t0 < 10.0 and all(p.poll() is not None for p in procs)
    procs = [subprocess.Popen(sleeper), subprocess.Popen(sleeper)]
    t0 = time.monotonic()
    assert bench.wait_ranks(procs, timeout_s=0.5) == 124
    assert time.monotonic() - t0 < 10.0 and all(p.poll() is not None for p in procs)
    ok = [subprocess.Popen([sys.executable, "-c", "pass"]) for _ in range(3)]
    assert bench.wait_ranks(ok, timeout_s=30.0) == 0


def test_asm_prefetch_guard_checks_the_four_one_slice_backward_instantiations():
    """__graft_entry__.check_asm_prefetch disassembles the built code object: the un-waited inline-asm kernel-argument
    prefetch of the one-slice backward instantiations (csrc/iqlhip_kernels.h) must have its eight destination SGPRs
    untouched on every path up to an s_waitcnt lgkmcnt(0).  It fails closed — exactly four instantiations are expected
    (a changed template signature or a missing pattern raises instead of passing silently)."""
    import __graft_entry__ as g
    g.build()
    rep = g.check_asm_prefetch()
    assert len(rep) == 4
    for name, r in rep.items():
        assert "ELb0EEv" in name and len(r["prefetch_dest_sgprs"]) == 8
        assert r["retiring_waits"] >= 1 and r["instructions_checked"] > 100
