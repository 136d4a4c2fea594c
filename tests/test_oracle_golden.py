"""Pins the oracle (oracle/iql_oracle.py, a closed-form numpy restatement) against
outputs of the reference captured by tools/make_goldens.py.  CPU only.

Tolerances: the reference runs in fp32, and so does the oracle; two fp32
evaluations of the same maths in different summation orders differ by ~1e-7
relative on the losses and up to ~1e-5 relative-to-max on gradient tensors
with heavy cancellation (observed: 1.1e-5 on one tensor), hence grad_rtol 3e-5 and — a first-step Adam moment being (1 - beta) * g — moment_rtol 5e-5 HERE (numpy's
summation order vs aten's; observed 1.08e-5 on one bias).  The HIP kernels are held to the tighter 1e-5 / 1e-5 in
tests/test_hip_parity.py (observed 2.6e-6 / 5.4e-6, profiles/r02_parity_margins.txt).
"""
import numpy as np
import pytest

import synth
from helpers import (ACT_CASES, FREERUN_CASES, SINGLE_STEP_CASES, act_case_params, assert_losses, batch_from,
                     check_state_against_golden, check_step_against_golden, load_golden, single_step_inputs, step_batch,
                     sub)
from oracle import iql_oracle as O


@pytest.mark.parametrize("name", SINGLE_STEP_CASES)
def test_single_step_matches_reference(name):
    z, meta = load_golden(name)
    params, batch, hyper = single_step_inputs(meta)
    opt = O.new_opt_state(params)
    newp, newo, info = O.iql_step(params, opt, batch, hyper, meta["lrs"])
    check_step_against_golden(z, meta, info, newp, newo, grad_rtol=3e-5, param_atol=2e-6,
                              loss_rtol=1e-5, target_atol=1e-7, moment_rtol=5e-5)


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
                              target_atol=1e-7, moment_rtol=5e-5)
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


def _copy(tree):
    return {n: {k: v.copy() for k, v in t.items()} for n, t in tree.items()}


@pytest.mark.parametrize("name", ["g11_resume_S17A6_gauss", "g11_resume_S29A8_det"])
def test_checkpoint_resume_matches_reference(name):
    """G11: n steps -> state_dict -> load_state_dict into a fresh trainer -> one more step, as the reference ran it
    (finetune/iql.py:565-593): after the load the target net is a COPY OF qf (the saved target is not in the
    checkpoint), optimiser moments / step counts / the cosine schedule continue."""
    z, meta = load_golden(name)
    S, A, B, T = meta["S"], meta["A"], meta["B"], meta["max_steps"]
    hyper = dict(meta["hyper"])
    hyper["deterministic"] = not meta["gaussian"]
    params = synth.synth_params(S, A, seed=meta["seed"], gaussian=meta["gaussian"])
    opt = O.new_opt_state(params)
    for k in range(meta["n_before"]):
        lrs = dict(meta["lrs"], pi=O.cosine_lr(meta["lrs"]["pi"], k, T))
        params, opt, info = O.iql_step(params, opt, step_batch(S, A, B, meta["batch_seed0"] + k), hyper, lrs)
        assert_losses([info["value_loss"], info["q_loss"], info["actor_loss"]], z["losses_before"][k], 1e-5, f"step {k}")
    check_state_against_golden(z, meta, "ckpt", params, opt, param_atol=4e-6, moment_rtol=3e-5, target_atol=1e-7)
    assert abs(O.cosine_lr(meta["lrs"]["pi"], meta["n_before"], T) - float(z["ckpt.actor_lr"][0])) <= 1e-15
    assert float(z["ckpt.q_step"][0]) == meta["n_before"] and meta["target_equals_qf_after_load"]
    # the fresh trainer's own parameters are overwritten by the load; target := qf
    loaded = _copy(params)
    loaded["qt1"], loaded["qt2"] = _copy({"x": params["q1"]})["x"], _copy({"x": params["q2"]})["x"]
    k = meta["n_before"]
    lrs = dict(meta["lrs"], pi=O.cosine_lr(meta["lrs"]["pi"], k, T))
    newp, newo, info = O.iql_step(loaded, opt, step_batch(S, A, B, meta["batch_seed0"] + k), hyper, lrs)
    assert_losses([info["value_loss"], info["q_loss"], info["actor_loss"]], z["losses_after"], 1e-5, "after load")
    check_state_against_golden(z, meta, "after", newp, newo, param_atol=4e-6, moment_rtol=3e-5, target_atol=1e-6)
    assert meta["total_it_after"] == k + 1


def test_jsrl_handoff_matches_reference():
    """G12: the reference's own jsrl_utils.get_learning_agent (make_actor -> partial_load_state_dict(guide.state_dict())
    -> total_it = offline_iterations, jsrl_utils.py:350-355) followed by the first online update, which samples 256
    rows with replacement from a 10 000-row ring holding ONE row (jsrl_w_iql.py:540-548)."""
    z, meta = load_golden("g12_jsrl_handoff_S29A8")
    S, A, B = meta["S"], meta["A"], meta["B"]
    hyper = dict(meta["hyper"], deterministic=False)
    params = synth.synth_params(S, A, seed=meta["seed"])
    opt = O.new_opt_state(params)
    for k in range(2):       # the guide's offline steps (cosine schedule over offline_iterations)
        lrs = dict(meta["lrs"], pi=O.cosine_lr(meta["lrs"]["pi"], k, meta["offline_iterations"]))
        b = step_batch(S, A, B, meta["guide_batch_seed0"] + k, p_done=0.001, antmaze_rewards=True)
        params, opt, info = O.iql_step(params, opt, b, hyper, lrs)
        assert_losses([info["value_loss"], info["q_loss"], info["actor_loss"]], z["guide_losses"][k], 1e-5)
    assert meta["total_it_after_handoff"] == meta["offline_iterations"] and not meta["learner_has_schedule"]
    assert meta["learner_opt_state_len"] == 0 and meta["params_equal_guide"] and meta["target_equals_guide_qf"]
    assert meta["all_curriculum_stages"] == [300.0] and meta["agent_type_stage"] == 1
    # the learner: the guide's networks, target := qf, FRESH optimisers, constant learning rates
    learner = _copy(params)
    learner["qt1"], learner["qt2"] = _copy({"x": params["q1"]})["x"], _copy({"x": params["q2"]})["x"]
    one = step_batch(S, A, 1, meta["row_seed"], antmaze_rewards=True)
    one["d"][:] = 0.0                                   # add_transition(..., done=False)
    np.random.seed(meta["np_seed"])
    idx = np.random.randint(0, 1, size=B)               # size 1 ring: every index is 0
    batch = {k: v[idx] for k, v in one.items()}
    newp, newo, info = O.iql_step(learner, O.new_opt_state(learner), batch, hyper, meta["lrs"])
    assert_losses([info["value_loss"], info["q_loss"], info["actor_loss"]], z["losses"], 1e-5)
    check_step_against_golden(z, meta, {k: v for k, v in info.items() if k in ("value_loss", "q_loss", "actor_loss", "grads")},
                              newp, newo, grad_rtol=3e-5, param_atol=2e-6, loss_rtol=1e-5, target_atol=1e-7,
                              moment_rtol=5e-5)
    assert meta["total_it_after_step"] == meta["offline_iterations"] + 1


def test_config5_share_with_dropout_matches_reference():
    """G14: configs[4]'s per-GPU share (obs 39, act 28, 1024 rows, actor dropout 0.1 by mask injection) in fp32."""
    z, meta = load_golden("g14_c5_B1024_dropout")
    params, batch, hyper = single_step_inputs(meta)
    p = meta["dropout"]
    k0, k1 = synth.synth_dropout_keep(meta["B"], p, seed=meta["seed"])
    masks = (k0.astype(np.float32) / np.float32(1.0 - p), k1.astype(np.float32) / np.float32(1.0 - p))
    newp, newo, info = O.iql_step(params, O.new_opt_state(params), batch, hyper, meta["lrs"], actor_masks=masks)
    info2 = {k: v for k, v in info.items() if k not in ("next_v", "target_q", "adv")}
    check_step_against_golden(z, meta, info2, newp, newo, grad_rtol=3e-5, param_atol=2e-6, loss_rtol=1e-5,
                              target_atol=1e-7, moment_rtol=5e-5)
