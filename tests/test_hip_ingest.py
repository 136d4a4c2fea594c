"""GPU tests of the device-side dataset ingest and of the offline-flavour classes on the GPU."""
import numpy as np
import pytest
import torch

import synth
from helpers import assert_losses, check_step_against_golden, load_golden, single_step_inputs

pytestmark = pytest.mark.gpu


def test_device_mean_std_and_normalisation_against_numpy():
    """compute_mean_std / normalize_states (finetune/iql.py:77-84) on the device.  Stated tolerance: the float64-
    accumulated device reduction agrees with a float64 numpy evaluation to 1e-6 relative (it is the more exact of the
    two) and with numpy's own float32 result — what the reference computes — to 1e-3 relative at this size: numpy reduces
    axis 0 of a float32 array by adding the rows one after the other in float32, so ITS error grows with the row count
    (observed here, 1 M rows with offsets up to 50: 1.8e-4 of |mean| + std, while the device result sits at < 1e-6 of
    the float64 value).  The normalisation is bit-identical to numpy's given the same mean / std."""
    import iql
    S, A, N = 17, 6, 1_000_003
    rng = np.random.default_rng(5)
    data = synth.synth_transitions(N, S, A, seed=9)
    # un-normalised, differently scaled columns with large offsets (what raw D4RL observations look like)
    scale = rng.uniform(0.05, 30.0, size=S).astype(np.float32)
    shift = rng.uniform(-50.0, 50.0, size=S).astype(np.float32)
    data["observations"] = data["observations"] * scale + shift
    data["next_observations"] = data["next_observations"] * scale + shift
    eps = 1e-3
    m32, s32 = iql.compute_mean_std(data["observations"], eps)            # the reference's numpy form (float32)
    m64 = data["observations"].astype(np.float64).mean(0)
    s64 = data["observations"].astype(np.float64).std(0) + eps
    buf = iql.ReplayBuffer(S, A, N + 10, "cuda")
    buf.load_d4rl_dataset(data)
    mean, std = buf.state_mean_std(eps)
    assert mean.dtype == np.float32 and std.shape == (S,)
    assert np.max(np.abs(mean - m64) / (np.abs(m64) + s64)) <= 1e-6
    assert np.max(np.abs(std - s64) / s64) <= 1e-6
    assert np.max(np.abs(mean - m32) / (np.abs(m32) + s32)) <= 1e-3 and np.max(np.abs(std - s32) / s32) <= 1e-3
    again = buf.state_mean_std(eps)
    assert np.array_equal(again[0], mean) and np.array_equal(again[1], std)          # deterministic
    # in-place normalisation with numpy's own mean / std: bit-identical rows
    buf.normalize_states_(m32, s32)
    want_s = iql.normalize_states(data["observations"], m32, s32)
    want_ns = iql.normalize_states(data["next_observations"], m32, s32)
    idx = torch.tensor([0, 1, 12345, N - 1], device="cuda")
    s, a, r, ns, d = buf.gather(idx)
    ii = idx.cpu().numpy()
    assert np.array_equal(s.cpu().numpy(), want_s[ii]) and np.array_equal(ns.cpu().numpy(), want_ns[ii])
    assert np.array_equal(a.cpu().numpy(), data["actions"][ii]) and np.array_equal(r.cpu().numpy()[:, 0], data["rewards"][ii])
    full = buf._rows[:N].cpu().numpy()
    assert np.array_equal(full[:, :S], want_s) and np.array_equal(full[:, S + A: 2 * S + A], want_ns)
    assert float(buf._rows[N:].abs().max()) == 0.0                                   # rows beyond size untouched
    # host-array entry point
    import iqlhip_ingest as ing
    m2, s2 = ing.compute_mean_std_device(data["observations"], eps)
    assert np.array_equal(m2, mean) and np.array_equal(s2, std)


def test_offline_flavour_step_matches_reference_fixture():
    """The offline classes (iql_offline.py): a policy built with dropout=0.0 carries nn.Dropout(0.0) layers (keys
    net.{0,3,6}) under the offline gate; its step is the same arithmetic — fixture g1_S17A6_gauss_b3."""
    import iql_offline as off
    from hip_helpers import _load_mlp, read_moments, read_params, to_torch_batch
    z, meta = load_golden("g1_S17A6_gauss_b3")
    params, batch, hyper = single_step_inputs(meta)
    S, A = meta["S"], meta["A"]
    qf, vf, actor = off.TwinQ(S, A), off.ValueFunction(S), off.GaussianPolicy(S, A, 1.0, dropout=0.0)
    assert list(actor.state_dict().keys())[1] == "net.net.0.weight" and "net.net.3.weight" in actor.state_dict()
    _load_mlp(vf.v, params["vf"]); _load_mlp(qf.q1, params["q1"]); _load_mlp(qf.q2, params["q2"]); _load_mlp(actor.net, params["pi"])
    with torch.no_grad():
        actor.log_std.copy_(torch.from_numpy(params["pi"]["log_std"]))
    qf, vf, actor = qf.cuda(), vf.cuda(), actor.cuda()
    tr = off.ImplicitQLearning(1.0, actor, torch.optim.Adam(actor.parameters(), lr=3e-4), qf,
                               torch.optim.Adam(qf.parameters(), lr=3e-4), vf, torch.optim.Adam(vf.parameters(), lr=3e-4),
                               iql_tau=hyper["iql_tau"], beta=hyper["beta"], max_steps=meta["max_steps"],
                               discount=hyper["discount"], tau=hyper["tau"], device="cuda")
    _load_mlp(tr.q_target.q1, params["qt1"]); _load_mlp(tr.q_target.q2, params["qt2"])
    log = tr.train(to_torch_batch(batch))
    assert_losses([log["value_loss"], log["q_loss"], log["actor_loss"]], z["losses"], 1e-5)
    check_step_against_golden(z, meta, None, read_params(tr), read_moments(tr))
    assert "actor_lr_schedule" in tr.state_dict() and tr.state_dict()["actor_lr_schedule"]["last_epoch"] == 1


def test_online_step_equals_add_sample_train():
    """ImplicitQLearning.online_step (one library call: ring write + gather from pinned host words + step) against the
    reference's three calls add_transition -> sample -> train (finetune/iql.py:741-773) on a ring that wraps: same
    numpy index draws, same losses, same parameters and ring contents — bit for bit."""
    import iql
    from hip_helpers import build_hip_trainer, read_params
    S, A, B, cap = 29, 8, 256, 50
    params = synth.synth_params(S, A, seed=141)
    hyper = {"iql_tau": 0.9, "beta": 10.0, "discount": 0.99, "tau": 0.005}
    lrs = {"v": 3e-4, "q": 3e-4, "pi": 3e-4}
    tr_a = build_hip_trainer(params, S, A, True, hyper, lrs, 1000)
    tr_b = build_hip_trainer(params, S, A, True, hyper, lrs, 1000)
    buf_a, buf_b = iql.ReplayBuffer(S, A, cap, "cuda"), iql.ReplayBuffer(S, A, cap, "cuda")
    stream = synth.synth_transitions(130, S, A, seed=142, antmaze_rewards=True)
    np.random.seed(5)
    logs_a = []
    for i in range(130):                 # the ring (50 rows) wraps twice; the first draws come from a 1-row buffer
        buf_a.add_transition(stream["observations"][i], stream["actions"][i], float(stream["rewards"][i]),
                             stream["next_observations"][i], bool(stream["terminals"][i]))
        logs_a.append(tr_a.train(buf_a.sample(B)))
    np.random.seed(5)
    for i in range(130):
        log = tr_b.online_step(buf_b, stream["observations"][i], stream["actions"][i], float(stream["rewards"][i]),
                               stream["next_observations"][i], bool(stream["terminals"][i]), B)
        assert log == logs_a[i], i
    assert (buf_b._pointer, buf_b._size) == (buf_a._pointer, buf_a._size) == (130 % cap, cap)
    assert torch.equal(buf_a._rows, buf_b._rows)
    pa, pb = read_params(tr_a), read_params(tr_b)
    for n in pa:
        for k in pa[n]:
            assert np.array_equal(pa[n][k], pb[n][k]), (n, k)
    assert tr_b.total_it == 130 and tr_a.actor_optimizer.param_groups[0]["lr"] == tr_b.actor_optimizer.param_groups[0]["lr"]
    with pytest.raises(IndexError):      # an index outside the ring is refused on the host like the reference's indexing (a gather would fault)
        import ctypes as C
        import iqlhip_binding as hb
        bad = np.full(B, cap, dtype=np.int64)
        row = np.zeros(buf_b._ld, dtype=np.float32)
        sc = hb.StepScalars()
        tr_b._fill_scalars(sc, {"v": 1, "q": 1, "pi": 1}, tr_b._current_lrs(), 1.0 / B)
        out = (C.c_float * 3)()
        hb.check(hb.lib().iqlhip_online_step(tr_b._ctx, buf_b._rows.data_ptr(), buf_b._ld, cap, 0, row.ctypes.data,
                                             bad.ctypes.data, B, C.byref(sc), out, None, 1.0, 0, None, tr_b._stream()))
    # act_next: the next iteration's act() under the same synchronisation == a separate actor.act() after the step
    for mode in ("eval", "train"):
        getattr(tr_a.actor, mode)()
        getattr(tr_b.actor, mode)()
        np.random.seed(9)
        buf_a.add_transition(stream["observations"][0], stream["actions"][0], -1.0, stream["next_observations"][0], False)
        la = tr_a.train(buf_a.sample(B))
        aa = tr_a.actor.act(stream["next_observations"][0], "cuda")
        np.random.seed(9)
        lb, ab = tr_b.online_step(buf_b, stream["observations"][0], stream["actions"][0], -1.0,
                                  stream["next_observations"][0], False, B, act_next=stream["next_observations"][0])
        assert la == lb and ab.shape == (A,)
        if mode == "eval":
            assert np.array_equal(aa, ab)            # the mean action: same kernels, same weights
        else:                                        # training mode: each trainer draws from its own Philox stream position
            assert np.all(np.abs(ab) <= 1.0) and tr_b._ctx is not None


def test_batch_growth_keeps_random_stream_positions_and_survives_a_refused_size():
    """Growing the batch re-creates the library context: the dropout-mask and act()-noise stream positions carry over
    (the streams do not replay from their beginning), and a size the library refuses leaves the trainer usable."""
    import ctypes as C
    import iqlhip_binding as hb
    from hip_helpers import build_hip_trainer, to_torch_batch
    S, A = 17, 6
    params = synth.synth_params(S, A, seed=151)
    hyper = {"iql_tau": 0.7, "beta": 3.0, "discount": 0.99, "tau": 0.005}
    tr = build_hip_trainer(params, S, A, True, hyper, {"v": 3e-4, "q": 3e-4, "pi": 3e-4}, 1000, dropout=0.1)

    def batch(B, seed):
        d = synth.synth_transitions(B, S, A, seed=seed)
        return to_torch_batch({"s": d["observations"], "a": d["actions"], "r": d["rewards"], "ns": d["next_observations"],
                               "d": d["terminals"]})

    for k in range(3):
        tr.train(batch(256, k))
    tr.actor.act(np.zeros(S, dtype=np.float32), "cuda")      # eval/train-mode call counts are the library's own
    ctr = (C.c_uint64 * 2)()
    hb.check(hb.lib().iqlhip_get_counters(tr._ctx, ctr))
    before = (int(ctr[0]), int(ctr[1]))
    assert before[0] == 3
    tr.train(batch(512, 9))                                   # grows: a new context behind the same trainer
    assert tr._max_batch == 512
    hb.check(hb.lib().iqlhip_get_counters(tr._ctx, ctr))
    assert int(ctr[0]) == before[0] + 1 and int(ctr[1]) == before[1]
    with pytest.raises(ValueError):
        tr.reserve_batch(1 << 20)                             # beyond the library's range: refused up front
    log = tr.train(batch(256, 10))                            # ... and the existing context still trains
    assert all(np.isfinite(v) for v in log.values())
