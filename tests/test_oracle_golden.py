"""Pins the oracle (oracle/iql_oracle.py, a closed-form numpy restatement) against
outputs of the reference captured by tools/make_goldens.py.  CPU only.

Tolerances: the reference runs in fp32, and so does the oracle; two fp32
evaluations of the same maths in different summation orders differ by ~1e-7
relative on the losses and up to ~1e-5 relative-to-max on gradient tensors
with heavy cancellation (observed: 1.1e-5 on one tensor), hence grad_rtol 3e-5.
"""
import numpy as np
import pytest

import synth
from helpers import (ACT_CASES, FREERUN_CASES, SINGLE_STEP_CASES, act_case_params, assert_losses, batch_from,
                     check_step_against_golden, load_golden, single_step_inputs, sub)
from oracle import iql_oracle as O


@pytest.mark.parametrize("name", SINGLE_STEP_CASES)
def test_single_step_matches_reference(name):
    z, meta = load_golden(name)
    params, batch, hyper = single_step_inputs(meta)
    opt = O.new_opt_state(params)
    newp, newo, info = O.iql_step(params, opt, batch, hyper, meta["lrs"])
    check_step_against_golden(z, meta, info, newp, newo, grad_rtol=3e-5, param_atol=2e-6,
                              loss_rtol=1e-5, target_atol=1e-7)


@pytest.mark.parametrize("name", FREERUN_CASES)
def test_free_run_matches_reference(name):
    z, meta = load_golden(name)
    S, A = meta["S"], meta["A"]
    params = synth.synth_params(S, A, seed=meta["seed"], gaussian=meta["gaussian"])
    data = synth.synth_transitions(meta["N"], S, A, seed=2000 + meta["seed"])
    hyper = dict(meta["hyper"])
    hyper["deterministic"] = not meta["gaussian"]
    opt = O.new_opt_state(params)
    for k in range(meta["n_steps"]):
        lr_pi = O.cosine_lr(meta["lrs"]["pi"], k, meta["max_steps"])
        assert abs(lr_pi - z["actor_lr_used"][k]) <= 1e-12 * meta["lrs"]["pi"] + 1e-18
        lrs = {"v": meta["lrs"]["v"], "q": meta["lrs"]["q"], "pi": float(z["actor_lr_used"][k])}
        params, opt, info = O.iql_step(params, opt, batch_from(data, z["indices"][k]), hyper, lrs)
        assert_losses([info["value_loss"], info["q_loss"], info["actor_loss"]], z["losses"][k], 1e-5,
                      what=f"step {k}")
    # after 10 free-running steps parameters agree to ~1e-6 (Adam amplifies rounding of tiny grads)
    for net, tensors in params.items():
        for t, p in tensors.items():
            want = z[f"param.{net}.{t}"]
            got = sub(p, meta["stride"]).reshape(want.shape)
            assert np.max(np.abs(got - want)) <= 2e-5, (net, t)


def test_gather_matches_reference():
    z, meta = load_golden("g3_gather")
    data = synth.synth_transitions(meta["N"], meta["S"], meta["A"], seed=meta["data_seed"])
    np.random.seed(meta["np_seed"])
    idx = np.random.randint(0, meta["size"], size=meta["B"])
    assert np.array_equal(idx, z["indices"])
    s, a, r, ns, d = O.replay_sample(data, idx)
    for got, key in ((s, "s"), (a, "a"), (r, "r"), (ns, "ns"), (d, "d")):
        assert got.shape == z[key].shape
        assert np.array_equal(got, z[key]), key


def test_cosine_lr_closed_form_matches_reference_recursion():
    z, meta = load_golden("g5_lr")
    lrs = z["lrs"]
    T = meta["T"]
    for t in range(len(lrs)):
        want = lrs[t]
        got = O.cosine_lr(meta["base_lr"], t, T)
        assert abs(got - want) <= 1e-12 * meta["base_lr"] + 1e-19, (t, got, want)
    assert meta["no_schedule_is_none"] is True


def test_dp_shard_sum_equals_big_batch():
    """G8: summing 8 shard gradients (each scaled by the GLOBAL batch) == the B=2048 gradient."""
    z, meta = load_golden("g8_dp_B2048")
    params, batch, hyper = single_step_inputs(meta)
    B, W = meta["B"], 8
    b = B // W
    acc, losses = None, np.zeros(3)
    for r in range(W):
        sl = slice(r * b, (r + 1) * b)
        shard = {k: v[sl] for k, v in batch.items()}
        info = O.iql_losses_and_grads(params, shard, hyper, grad_scale_rows=B)
        losses += np.array([info["value_loss"], info["q_loss"], info["actor_loss"]], dtype=np.float64) / W
        if acc is None:
            acc = {n: {k: g.astype(np.float64) for k, g in t.items()} for n, t in info["grads"].items()}
        else:
            for n, t in info["grads"].items():
                for k, g in t.items():
                    acc[n][k] += g
    assert_losses(losses, z["losses"], 1e-5)
    for n, t in acc.items():
        for k, g in t.items():
            want = z[f"grad.{n}.{k}"]
            got = sub(g, meta["stride"]).reshape(want.shape)
            assert np.max(np.abs(got - want)) <= 3e-5 * np.max(np.abs(want)), (n, k)


def test_fp64_oracle_brackets_reference():
    """The reference (fp32) sits within fp32 rounding of an fp64 evaluation of the same step."""
    z, meta = load_golden("g1_S17A6_gauss_b3")
    params, batch, hyper = single_step_inputs(meta)
    info = O.iql_losses_and_grads(params, batch, hyper, dtype=np.float64)
    assert_losses([info["value_loss"], info["q_loss"], info["actor_loss"]], z["losses"], 2e-6)


@pytest.mark.parametrize("name", ["g1_S17A6_gauss_b3", "g1_S29A8_det_b10", "g7_edge_gauss"])
def test_torch_port_matches_reference(name):
    """The PyTorch-CPU port used as bench.py's cpu_baseline reproduces the reference's losses."""
    import torch
    from oracle import iql_torch_port as port
    z, meta = load_golden(name)
    params, batch, hyper = single_step_inputs(meta)
    torch.set_num_threads(1)
    tr = port.CpuIQL(meta["S"], meta["A"], params=params, gaussian=meta["gaussian"], iql_tau=hyper["iql_tau"],
                     beta=hyper["beta"], discount=hyper["discount"], tau=hyper["tau"], lrs=meta["lrs"],
                     max_steps=meta["max_steps"])
    tb = [torch.from_numpy(batch["s"]), torch.from_numpy(batch["a"]), torch.from_numpy(batch["r"][:, None].copy()),
          torch.from_numpy(batch["ns"]), torch.from_numpy(batch["d"][:, None].copy())]
    log = tr.train(tb)
    assert_losses([log["value_loss"], log["q_loss"], log["actor_loss"]], z["losses"], 1e-6)
    want = z["param.q1.w1"]
    got = sub(tr.qf.q1.net[2].weight.detach().numpy(), meta["stride"])
    assert np.max(np.abs(got - want)) <= 1e-7


@pytest.mark.parametrize("name", ["g9_dropout_S39A28_gauss", "g9_dropout_S17A6_det"])
def test_dropout_step_with_injected_masks_matches_reference(name):
    """Actor dropout: the reference ran with the same keep-masks injected into nn.Dropout."""
    z, meta = load_golden(name)
    params, batch, hyper = single_step_inputs(meta)
    p = meta["dropout"]
    k0, k1 = synth.synth_dropout_keep(meta["B"], p, seed=meta["seed"])
    masks = (k0.astype(np.float32) / np.float32(1.0 - p), k1.astype(np.float32) / np.float32(1.0 - p))
    newp, newo, info = O.iql_step(params, O.new_opt_state(params), batch, hyper, meta["lrs"], actor_masks=masks)
    info2 = {k: v for k, v in info.items() if k not in ("next_v", "target_q", "adv")}
    check_step_against_golden(z, meta, info2, newp, newo, grad_rtol=3e-5, param_atol=2e-6, loss_rtol=1e-5,
                              target_atol=1e-7)
    bits = synth.pack_keep_bits(k0)
    assert bits.shape == (meta["B"], 8) and ((bits[3, 1] >> 5) & 1) == int(k0[3, 37])


@pytest.mark.parametrize("name", ACT_CASES)
def test_actor_act_matches_reference(name):
    """G10: the oracle's act() against GaussianPolicy.act / DeterministicPolicy.act of the reference (eval mode, one
    state per call) and against the training-mode formula with fixture noise."""
    z, meta = load_golden(name)
    pi = act_case_params(meta, z)
    got = O.actor_act(pi, z["states"], meta["max_action"])
    assert got.shape == z["actions_eval"].shape
    assert np.max(np.abs(got - z["actions_eval"])) <= 2e-6 * max(1.0, meta["max_action"])
    assert np.max(np.abs(got)) <= meta["max_action"]
    if meta["gaussian"]:
        gn = O.actor_act(pi, z["states"], meta["max_action"], noise=z["noise"])
        assert np.max(np.abs(gn - z["actions_noise"])) <= 2e-6 * max(1.0, meta["max_action"])
        assert np.any(np.abs(gn) == meta["max_action"])        # the clamp is exercised (log_std[0] = 3 -> sigma = e^2)
