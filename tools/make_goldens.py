#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in the build container.

    python tools/make_goldens.py [--ref /root/reference] [--out tests/golden]

The reference (LaurenYTaylor/jsrl-CORL, algorithms/finetune/iql.py) is imported
from where it lies; nothing of it is copied.  Its host-only third-party imports
that are not installed here (gym, gymnasium, d4rl, pyrallis, wandb) are replaced
by empty stub modules — the classes on the IQL step path use torch+numpy only
(SURVEY.md Appendix B).  Inputs come from jsrl-corl_amd/synth.py (numpy PCG64),
so a fixture holds (seed, dims, hyper-parameters) + the reference's OUTPUTS.

Large tensors are stored sub-sampled (every `stride`-th element of the
flattened tensor); tests apply the same sub-sampling to what they compare.
This script never runs on the GPU box (the reference does not travel).
"""
from __future__ import annotations

import argparse
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _load_synth():
    spec = importlib.util.spec_from_file_location("iqlhip_synth", os.path.join(ROOT, "jsrl-corl_amd", "synth.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


synth = _load_synth()


def import_reference(ref_root: str):
    """Import the reference's finetune/iql.py with stubbed host-only deps."""
    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class _Env:  # annotation target only
        pass

    for pkg in ("gym", "gymnasium"):
        wr = stub(pkg + ".wrappers")
        sp = stub(pkg + ".spaces", Discrete=type("Discrete", (), {}))
        stub(pkg, Env=_Env, wrappers=wr, spaces=sp, register_envs=lambda *a, **k: None)
    stub("d4rl")
    stub("wandb")
    stub("pyrallis", wrap=lambda *a, **k: (lambda fn: fn), dump=lambda *a, **k: None,
         parse=lambda *a, **k: None)
    sys.path.insert(0, os.path.join(ref_root, "algorithms", "finetune"))
    import iql as ref_iql  # noqa: E402  (the reference module)
    return ref_iql


# ----------------------------------------------------------------------------
def sub(arr: np.ndarray, stride: int) -> np.ndarray:
    a = np.asarray(arr)
    if a.size > 4096:
        return a.ravel()[::stride].copy()
    return a.copy()


NET_KEYS = {  # our tensor names -> reference state_dict keys
    "vf": {"w0": "v.net.0.weight", "b0": "v.net.0.bias", "w1": "v.net.2.weight", "b1": "v.net.2.bias",
           "w2": "v.net.4.weight", "b2": "v.net.4.bias"},
    "q1": {"w0": "q1.net.0.weight", "b0": "q1.net.0.bias", "w1": "q1.net.2.weight", "b1": "q1.net.2.bias",
           "w2": "q1.net.4.weight", "b2": "q1.net.4.bias"},
    "q2": {"w0": "q2.net.0.weight", "b0": "q2.net.0.bias", "w1": "q2.net.2.weight", "b1": "q2.net.2.bias",
           "w2": "q2.net.4.weight", "b2": "q2.net.4.bias"},
    "pi": {"w0": "net.net.0.weight", "b0": "net.net.0.bias", "w1": "net.net.2.weight", "b1": "net.net.2.bias",
           "w2": "net.net.4.weight", "b2": "net.net.4.bias", "log_std": "log_std"},
}


def build_trainer(ref, S, A, params, gaussian, hyper, lrs, max_steps, dropout=0.0):
    qf = ref.TwinQ(S, A)
    vf = ref.ValueFunction(S)
    actor = (ref.GaussianPolicy if gaussian else ref.DeterministicPolicy)(S, A, 1.0, dropout=dropout)
    mods = {"vf": vf, "q1": qf, "q2": qf, "pi": actor}
    NET_KEYS["pi"] = _pi_keys(actor)
    with torch.no_grad():
        for net, keys in NET_KEYS.items():
            sd = dict(mods[net].named_parameters())
            for ours, theirs in keys.items():
                if ours in params[net]:
                    sd[theirs].copy_(torch.from_numpy(params[net][ours]))
    v_opt = torch.optim.Adam(vf.parameters(), lr=lrs["v"])
    q_opt = torch.optim.Adam(qf.parameters(), lr=lrs["q"])
    a_opt = torch.optim.Adam(actor.parameters(), lr=lrs["pi"])
    tr = ref.ImplicitQLearning(
        max_action=1.0, actor=actor, actor_optimizer=a_opt, q_network=qf, q_optimizer=q_opt,
        v_network=vf, v_optimizer=v_opt, iql_tau=hyper["iql_tau"], beta=hyper["beta"],
        max_steps=max_steps, discount=hyper["discount"], tau=hyper["tau"], device="cpu")
    with torch.no_grad():  # distinct target weights
        tsd = dict(tr.q_target.named_parameters())
        for net, tnet in (("q1", "qt1"), ("q2", "qt2")):
            for ours, theirs in NET_KEYS[net].items():
                tsd[theirs].copy_(torch.from_numpy(params[tnet][ours]))
    return tr


def _pi_keys(actor):
    """state_dict keys of the actor's three Linear layers (net.net.{0,2,4} or {0,3,6} with dropout)."""
    idx = [i for i, m in enumerate(actor.net.net) if isinstance(m, torch.nn.Linear)]
    k = {}
    for j, name in enumerate(("0", "1", "2")):
        k["w" + name] = f"net.net.{idx[j]}.weight"
        k["b" + name] = f"net.net.{idx[j]}.bias"
    k["log_std"] = "log_std"
    return k


def grab(tr, stride, gaussian):
    """Current params / grads / Adam moments / target as sub-sampled numpy."""
    out = {}
    NET_KEYS["pi"] = _pi_keys(tr.actor)
    mods = {"vf": (tr.vf, tr.v_optimizer), "q1": (tr.qf, tr.q_optimizer), "q2": (tr.qf, tr.q_optimizer),
            "pi": (tr.actor, tr.actor_optimizer)}
    for net, keys in NET_KEYS.items():
        mod, opt = mods[net]
        named = dict(mod.named_parameters())
        for ours, theirs in keys.items():
            if ours == "log_std" and not gaussian:
                continue
            p = named[theirs]
            out[f"param.{net}.{ours}"] = sub(p.detach().numpy(), stride)
            if p.grad is not None:
                out[f"grad.{net}.{ours}"] = sub(p.grad.numpy(), stride)
            st = opt.state.get(p, None)
            if st:
                out[f"m.{net}.{ours}"] = sub(st["exp_avg"].numpy(), stride)
                out[f"v.{net}.{ours}"] = sub(st["exp_avg_sq"].numpy(), stride)
    tnamed = dict(tr.q_target.named_parameters())
    for net, tnet in (("q1", "qt1"), ("q2", "qt2")):
        for ours, theirs in NET_KEYS[net].items():
            out[f"param.{tnet}.{ours}"] = sub(tnamed[theirs].detach().numpy(), stride)
    return out


def to_batch(d, idx=None):
    if idx is None:
        idx = np.arange(d["observations"].shape[0])
    return [torch.from_numpy(d["observations"][idx]), torch.from_numpy(d["actions"][idx]),
            torch.from_numpy(d["rewards"][idx][:, None]), torch.from_numpy(d["next_observations"][idx]),
            torch.from_numpy(d["terminals"][idx][:, None])]


def ref_intermediates(tr, batch):
    """next_v / target_q / v / adv from PRE-step params, using the reference modules."""
    with torch.no_grad():
        s, a, r, ns, d = batch
        nv = tr.vf(ns)
        tq = tr.q_target(s, a)
        v = tr.vf(s)
        q1, q2 = tr.qf.both(s, a)
        out = {"next_v": nv.numpy(), "target_q": tq.numpy(), "v": v.numpy(), "adv": (tq - v).numpy(),
               "q1": q1.numpy(), "q2": q2.numpy()}
        pol = tr.actor(s)
        out["mu"] = (pol.mean if hasattr(pol, "mean") and not torch.is_tensor(pol) else pol).numpy()
    return out


def single_step_case(ref, name, S, A, gaussian, beta, iql_tau, B, seed, stride, outdir, edge=False):
    hyper = {"iql_tau": iql_tau, "beta": beta, "discount": 0.99, "tau": 0.005}
    lrs = {"v": 3e-4, "q": 3e-4, "pi": 3e-4}
    params = synth.synth_params(S, A, seed=seed, gaussian=gaussian)
    data = synth.synth_transitions(B, S, A, seed=1000 + seed)
    if edge:
        # G7: exp(beta*adv) overflow on some rows, log_std outside / on the clamp
        # bounds, terminal rows, duplicated rows.
        params["qt1"]["b2"] = params["qt1"]["b2"] + np.float32(60.0)
        params["qt2"]["b2"] = params["qt2"]["b2"] + np.float32(60.0)
        if gaussian:
            ls = np.zeros(A, dtype=np.float32)
            ls[:6] = np.array([3.0, -25.0, 0.5, 2.0, -20.0, -1.0], dtype=np.float32)[: min(6, A)]
            params["pi"]["log_std"] = ls
        data["terminals"][: B // 4] = 1.0
        for k in data:
            data[k][B // 2: B // 2 + 16] = data[k][:16]
        # half of the rows get a very negative advantage instead (w -> 0)
        data["observations"][B // 2 + 16:] *= np.float32(1.0)
    tr = build_trainer(ref, S, A, params, gaussian, hyper, lrs, max_steps=1000)
    batch = to_batch(data)
    inter = ref_intermediates(tr, batch)
    log = tr.train(batch)
    out = grab(tr, stride, gaussian)
    out.update({f"inter.{k}": v for k, v in inter.items()})
    out["losses"] = np.array([log["value_loss"], log["q_loss"], log["actor_loss"]], dtype=np.float64)
    out["lr_after"] = np.array([tr.actor_optimizer.param_groups[0]["lr"]], dtype=np.float64)
    meta = {"kind": "single_step", "S": S, "A": A, "gaussian": gaussian, "B": B, "seed": seed,
            "stride": stride, "hyper": hyper, "lrs": lrs, "max_steps": 1000, "edge": edge}
    out["meta"] = np.array(json.dumps(meta))
    np.savez(os.path.join(outdir, name + ".npz"), **out)
    print(f"{name}: losses={out['losses']}")


def dropout_case(ref, name, S, A, gaussian, pdrop, B, seed, stride, outdir):
    """Single step with actor dropout.  torch's dropout RNG stream cannot be matched by another
    implementation, so nn.Dropout.forward is replaced (for this call only) by a multiplication with
    keep-masks drawn from synth.synth_dropout_keep — the reference's MLP / loss / optimiser code runs
    unmodified, only the random mask is injected."""
    hyper = {"iql_tau": 0.8, "beta": 3.0, "discount": 0.99, "tau": 0.005}
    lrs = {"v": 3e-4, "q": 3e-4, "pi": 3e-4}
    params = synth.synth_params(S, A, seed=seed, gaussian=gaussian)
    data = synth.synth_transitions(B, S, A, seed=1000 + seed)
    tr = build_trainer(ref, S, A, params, gaussian, hyper, lrs, max_steps=1000, dropout=pdrop)
    n_drop = sum(isinstance(m, torch.nn.Dropout) for m in tr.actor.modules())
    assert n_drop == 2, n_drop
    k0, k1 = synth.synth_dropout_keep(B, pdrop, seed=seed)
    queue = [torch.from_numpy(k0.astype(np.float32) / np.float32(1.0 - pdrop)),
             torch.from_numpy(k1.astype(np.float32) / np.float32(1.0 - pdrop))]
    orig = torch.nn.Dropout.forward

    def injected(self, x):
        assert self.training and abs(self.p - pdrop) < 1e-12
        return x * queue.pop(0)

    torch.nn.Dropout.forward = injected
    try:
        batch = to_batch(data)
        log = tr.train(batch)
    finally:
        torch.nn.Dropout.forward = orig
    assert not queue
    out = grab(tr, stride, gaussian)
    out["losses"] = np.array([log["value_loss"], log["q_loss"], log["actor_loss"]], dtype=np.float64)
    out["lr_after"] = np.array([tr.actor_optimizer.param_groups[0]["lr"]], dtype=np.float64)
    meta = {"kind": "single_step_dropout", "S": S, "A": A, "gaussian": gaussian, "B": B, "seed": seed,
            "stride": stride, "hyper": hyper, "lrs": lrs, "max_steps": 1000, "edge": False, "dropout": pdrop,
            "actor_state_keys": list(tr.actor.state_dict().keys())}
    out["meta"] = np.array(json.dumps(meta))
    np.savez(os.path.join(outdir, name + ".npz"), **out)
    print(f"{name}: losses={out['losses']}")


def freerun_case(ref, name, S, A, gaussian, n_steps, B, N, seed, stride, outdir, T=1000):
    hyper = {"iql_tau": 0.7, "beta": 3.0, "discount": 0.99, "tau": 0.005}
    lrs = {"v": 3e-4, "q": 3e-4, "pi": 3e-4}
    params = synth.synth_params(S, A, seed=seed, gaussian=gaussian)
    data = synth.synth_transitions(N, S, A, seed=2000 + seed)
    tr = build_trainer(ref, S, A, params, gaussian, hyper, lrs, max_steps=T)
    buf = ref.ReplayBuffer(S, A, N, "cpu")
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        buf.load_d4rl_dataset({k: v.copy() for k, v in data.items()})
    np.random.seed(seed)
    losses, lrs_used, idxs = [], [], []
    for _ in range(n_steps):
        st = np.random.get_state()
        batch = buf.sample(B)
        np.random.set_state(st)
        idxs.append(np.random.randint(0, N, size=B))  # same draw the buffer just made
        lrs_used.append(tr.actor_optimizer.param_groups[0]["lr"])
        log = tr.train(batch)
        losses.append([log["value_loss"], log["q_loss"], log["actor_loss"]])
    out = grab(tr, stride, gaussian)
    out["losses"] = np.array(losses, dtype=np.float64)
    out["actor_lr_used"] = np.array(lrs_used, dtype=np.float64)
    out["indices"] = np.array(idxs, dtype=np.int64)
    out["total_it"] = np.array([tr.total_it])
    meta = {"kind": "free_run", "S": S, "A": A, "gaussian": gaussian, "B": B, "N": N, "seed": seed,
            "stride": stride, "hyper": hyper, "lrs": lrs, "max_steps": T, "n_steps": n_steps}
    out["meta"] = np.array(json.dumps(meta))
    np.savez(os.path.join(outdir, name + ".npz"), **out)
    print(f"{name}: last losses={losses[-1]}")


def gather_case(ref, outdir):
    S, A, N, B = 17, 6, 4096, 256
    data = synth.synth_transitions(N, S, A, seed=3)
    import contextlib
    import io
    buf = ref.ReplayBuffer(S, A, N + 100, "cpu")
    with contextlib.redirect_stdout(io.StringIO()):
        buf.load_d4rl_dataset({k: v.copy() for k, v in data.items()})
    np.random.seed(123)
    s, a, r, ns, d = buf.sample(B)
    np.random.seed(123)
    idx = np.random.randint(0, N, size=B)
    np.savez(os.path.join(outdir, "g3_gather.npz"), indices=idx, s=s.numpy(), a=a.numpy(), r=r.numpy(),
             ns=ns.numpy(), d=d.numpy(),
             meta=np.array(json.dumps({"S": S, "A": A, "N": N, "B": B, "data_seed": 3, "np_seed": 123,
                                       "capacity": N + 100, "size": buf._size, "pointer": buf._pointer})))
    print("g3_gather ok", r.shape, d.shape)


def ring_case(ref, outdir):
    S, A, cap = 3, 2, 8
    data = synth.synth_transitions(5, S, A, seed=4)
    extra = synth.synth_transitions(7, S, A, seed=5)
    import contextlib
    import io
    buf = ref.ReplayBuffer(S, A, cap, "cpu")
    with contextlib.redirect_stdout(io.StringIO()):
        buf.load_d4rl_dataset({k: v.copy() for k, v in data.items()})
    trace = [(buf._pointer, buf._size)]
    for i in range(7):
        buf.add_transition(extra["observations"][i], extra["actions"][i], float(extra["rewards"][i]),
                           extra["next_observations"][i], bool(extra["terminals"][i] > 0.5 or i == 2))
        trace.append((buf._pointer, buf._size))
    # error behaviour of load_d4rl_dataset
    errs = {}
    try:
        buf.load_d4rl_dataset(data)
    except ValueError as e:
        errs["nonempty"] = str(e)
    try:
        ref.ReplayBuffer(S, A, 3, "cpu").load_d4rl_dataset(data)
    except ValueError as e:
        errs["too_small"] = str(e)
    np.savez(os.path.join(outdir, "g4_ring.npz"), states=buf._states.numpy(), actions=buf._actions.numpy(),
             rewards=buf._rewards.numpy(), next_states=buf._next_states.numpy(), dones=buf._dones.numpy(),
             trace=np.array(trace, dtype=np.int64),
             meta=np.array(json.dumps({"S": S, "A": A, "capacity": cap, "load_seed": 4, "extra_seed": 5,
                                       "errors": errs})))
    print("g4_ring ok", trace[-1], errs)


def lr_case(ref, outdir):
    T = 1000
    S, A = 3, 2
    params = synth.synth_params(S, A, seed=0)
    hyper = {"iql_tau": 0.7, "beta": 3.0, "discount": 0.99, "tau": 0.005}
    tr = build_trainer(ref, S, A, params, True, hyper, {"v": 3e-4, "q": 3e-4, "pi": 3e-4}, max_steps=T)
    lrs = [tr.actor_optimizer.param_groups[0]["lr"]]
    tr.actor_optimizer._opt_called = True  # silence the scheduler-order warning
    for _ in range(3 * T // 2):
        tr.actor_lr_schedule.step()
        lrs.append(tr.actor_optimizer.param_groups[0]["lr"])
    tr2 = build_trainer(ref, S, A, params, True, hyper, {"v": 3e-4, "q": 3e-4, "pi": 3e-4}, max_steps=None)
    np.savez(os.path.join(outdir, "g5_lr.npz"), lrs=np.array(lrs, dtype=np.float64),
             meta=np.array(json.dumps({"T": T, "base_lr": 3e-4, "no_schedule_is_none": tr2.actor_lr_schedule is None,
                                       "sched_state_keys": sorted(tr.actor_lr_schedule.state_dict().keys())})))
    print("g5_lr ok", lrs[0], lrs[T], lrs[-1])


def statedict_case(ref, outdir):
    S, A = 17, 6
    params = synth.synth_params(S, A, seed=6)
    hyper = {"iql_tau": 0.7, "beta": 3.0, "discount": 0.99, "tau": 0.005}
    tr = build_trainer(ref, S, A, params, True, hyper, {"v": 3e-4, "q": 3e-4, "pi": 3e-4}, max_steps=1000)
    data = synth.synth_transitions(64, S, A, seed=61)
    tr.train(to_batch(data))
    sd = tr.state_dict()
    desc = {}
    for k in ("qf", "vf", "actor"):
        desc[k] = {kk: list(vv.shape) for kk, vv in sd[k].items()}
    for k in ("q_optimizer", "v_optimizer", "actor_optimizer"):
        o = sd[k]
        desc[k] = {"state_keys": sorted(str(x) for x in o["state"].keys()),
                   "entry_keys": sorted(o["state"][0].keys()),
                   "step0": float(o["state"][0]["step"]),
                   "param_group_keys": sorted(o["param_groups"][0].keys()),
                   "params": o["param_groups"][0]["params"]}
    desc["actor_lr_schedule"] = {k: (v if isinstance(v, (int, float, bool, str)) else str(v))
                                 for k, v in sd["actor_lr_schedule"].items()}
    desc["total_it"] = sd["total_it"]
    desc["top_keys"] = list(sd.keys())
    # after load_state_dict the target equals qf (reference quirk, Appendix A)
    tr2 = build_trainer(ref, S, A, synth.synth_params(S, A, seed=7), True, hyper,
                        {"v": 3e-4, "q": 3e-4, "pi": 3e-4}, max_steps=1000)
    tr2.load_state_dict(sd)
    same = all(torch.equal(a, b) for a, b in zip(tr2.q_target.parameters(), tr2.qf.parameters()))
    desc["target_equals_qf_after_load"] = bool(same)
    desc["target_requires_grad_after_load"] = bool(next(tr2.q_target.parameters()).requires_grad)
    log = tr2.train(to_batch(synth.synth_transitions(64, S, A, seed=62)))
    np.savez(os.path.join(outdir, "g6_statedict.npz"),
             losses_after_load=np.array([log["value_loss"], log["q_loss"], log["actor_loss"]], dtype=np.float64),
             meta=np.array(json.dumps(desc)))
    print("g6_statedict ok", desc["top_keys"])


def act_case(ref, name, S, A, gaussian, max_action, seed, outdir, n=48):
    """G10: GaussianPolicy.act / DeterministicPolicy.act (finetune/iql.py:371-379, 404-413), one state at a time in
    eval mode, plus — Gaussian only — the clamp(max_action * (mean + std * noise)) that act() forms in training mode,
    with the noise drawn here (the reference draws it inside dist.sample(), which no fixture can pin)."""
    params = synth.synth_params(S, A, seed=seed, gaussian=gaussian)
    rng = np.random.default_rng(9000 + seed)
    states = (rng.standard_normal((n, S)) * np.where(np.arange(n)[:, None] % 4 == 3, 6.0, 1.0)).astype(np.float32)
    actor = (ref.GaussianPolicy if gaussian else ref.DeterministicPolicy)(S, A, max_action)
    keys = _pi_keys(actor)
    log_std = None
    if gaussian:
        log_std = rng.uniform(-1.0, 0.5, size=A).astype(np.float32)
        log_std[0] = 3.0            # beyond LOG_STD_MAX = 2 -> clamped
        if A > 1:
            log_std[1] = -25.0      # below LOG_STD_MIN = -20 -> clamped
        params["pi"]["log_std"] = log_std
    with torch.no_grad():
        sd = dict(actor.named_parameters())
        for ours, theirs in keys.items():
            if ours in params["pi"]:
                sd[theirs].copy_(torch.from_numpy(params["pi"][ours]))
    actor.eval()
    acts = np.stack([actor.act(states[i], "cpu") for i in range(n)]).astype(np.float32)
    out = {"states": states, "actions_eval": acts}
    if gaussian:
        noise = rng.standard_normal((n, A)).astype(np.float32)
        with torch.no_grad():
            dist = actor(torch.from_numpy(states))
            a = dist.mean + dist.stddev * torch.from_numpy(noise)
            out["actions_noise"] = torch.clamp(max_action * a, -max_action, max_action).numpy().astype(np.float32)
        out["noise"] = noise
        out["log_std"] = log_std
    meta = {"S": S, "A": A, "gaussian": gaussian, "max_action": max_action, "seed": seed, "n": n}
    np.savez(os.path.join(outdir, name + ".npz"), meta=json.dumps(meta), **out)
    print(name, "ok; |a|max", float(np.abs(acts).max()))


def import_reference_jsrl(ref_root: str):
    """The reference's jsrl_utils.py (after import_reference: its `from iql import ...` binds the reference's iql)."""
    def stub(name, **attrs):
        if name in sys.modules:
            return sys.modules[name]
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class _Any:
        def __init__(self, *a, **k):
            pass

    pol = stub("stable_baselines3.sac.policies", Actor=_Any)
    sac = stub("stable_baselines3.sac", policies=pol)
    stub("stable_baselines3", SAC=_Any, sac=sac)
    import jsrl_utils  # noqa: E402  (the reference module; sys.path already holds algorithms/finetune)
    return jsrl_utils


def _step_batch(S, A, B, seed):
    return to_batch(synth.synth_transitions(B, S, A, seed=seed))


def resume_case(ref, name, S, A, gaussian, seed, stride, outdir, n_before=3, B=256):
    """G11: n_before steps -> state_dict() -> load_state_dict into a FRESH trainer (other initial
    parameters) -> one more step.  Pins the checkpoint contents after training (params, moments, step counts, LR
    schedule) and the continuation — incl. the reference's quirk that the target net is re-created as a copy of qf
    (finetune/iql.py:581-593)."""
    hyper = {"iql_tau": 0.7, "beta": 3.0, "discount": 0.99, "tau": 0.005}
    lrs = {"v": 3e-4, "q": 3e-4, "pi": 3e-4}
    T = 50                                     # short cosine period: the restored scheduler state matters
    tr = build_trainer(ref, S, A, synth.synth_params(S, A, seed=seed, gaussian=gaussian), gaussian, hyper, lrs, T)
    losses = []
    for k in range(n_before):
        log = tr.train(_step_batch(S, A, B, 3000 + 10 * seed + k))
        losses.append([log["value_loss"], log["q_loss"], log["actor_loss"]])
    sd = tr.state_dict()
    out = {f"ckpt.{k}": v for k, v in grab(tr, stride, gaussian).items() if not k.startswith("grad.")}
    out["ckpt.actor_lr"] = np.array([tr.actor_optimizer.param_groups[0]["lr"]], dtype=np.float64)
    out["ckpt.q_step"] = np.array([float(sd["q_optimizer"]["state"][0]["step"])])
    tr2 = build_trainer(ref, S, A, synth.synth_params(S, A, seed=seed + 1, gaussian=gaussian), gaussian, hyper, lrs, T)
    tr2.load_state_dict(sd)
    same = all(torch.equal(a, b) for a, b in zip(tr2.q_target.parameters(), tr2.qf.parameters()))
    log = tr2.train(_step_batch(S, A, B, 3000 + 10 * seed + n_before))
    out.update({f"after.{k}": v for k, v in grab(tr2, stride, gaussian).items() if not k.startswith("grad.")})
    out["losses_before"] = np.array(losses, dtype=np.float64)
    out["losses_after"] = np.array([log["value_loss"], log["q_loss"], log["actor_loss"]], dtype=np.float64)
    out["after.actor_lr"] = np.array([tr2.actor_optimizer.param_groups[0]["lr"]], dtype=np.float64)
    meta = {"kind": "resume", "S": S, "A": A, "gaussian": gaussian, "B": B, "seed": seed, "stride": stride,
            "hyper": hyper, "lrs": lrs, "max_steps": T, "n_before": n_before, "batch_seed0": 3000 + 10 * seed,
            "target_equals_qf_after_load": bool(same), "total_it_after": int(tr2.total_it)}
    out["meta"] = np.array(json.dumps(meta))
    np.savez(os.path.join(outdir, name + ".npz"), **out)
    print(f"{name}: before={losses[-1]} after={out['losses_after']}")


class _JsrlCfg:
    """The JsrlTrainConfig fields jsrl_utils.make_actor / get_learning_agent / prepare_finetuning read."""
    device = "cpu"
    actor_dropout = 0.0
    iql_deterministic = False
    vf_lr = qf_lr = actor_lr = 3e-4
    discount, tau, beta, iql_tau = 0.99, 0.005, 10.0, 0.9          # antmaze values (configs[2])
    n_curriculum_stages, horizon_fn, no_agent_types, rolling_mean_n, tolerance = 1, "time_step", True, 5, 0.05
    guide_heuristic_fn = None
    offline_iterations = 300
    batch_size = 256


def jsrl_handoff_case(ref, jsrl, name, S, A, seed, stride, outdir):
    """G12: the offline -> online switch of the JSRL loop, run by the reference's own code:
    guide trainer (2 offline steps) -> jsrl_utils.get_learning_agent (make_actor, partial_load_state_dict of the guide's
    state_dict, total_it = offline_iterations; jsrl_utils.py:350-355) -> a fresh 10 000-row online buffer holding ONE
    transition -> sample(256) (256 copies of that row: the update gate `t >= batch_size` counts global iterations,
    jsrl_w_iql.py:540-548) -> train."""
    cfg = _JsrlCfg()
    hyper = {"iql_tau": cfg.iql_tau, "beta": cfg.beta, "discount": cfg.discount, "tau": cfg.tau}
    lrs = {"v": cfg.vf_lr, "q": cfg.qf_lr, "pi": cfg.actor_lr}
    guide = build_trainer(ref, S, A, synth.synth_params(S, A, seed=seed), True, hyper, lrs, cfg.offline_iterations)
    g_losses = []
    for k in range(2):
        log = guide.train(to_batch(synth.synth_transitions(cfg.batch_size, S, A, seed=4000 + k, p_done=0.001,
                                                           antmaze_rewards=True)))
        g_losses.append([log["value_loss"], log["q_loss"], log["actor_loss"]])
    torch.manual_seed(1234)        # make_actor's own random init is overwritten by the partial load (optimizers stay fresh)
    import contextlib
    import io
    trainer, cfg2 = jsrl.get_learning_agent(cfg, guide, 300, S, A, 1.0)
    desc = {"total_it_after_handoff": int(trainer.total_it), "learner_has_schedule": trainer.actor_lr_schedule is not None,
            "learner_opt_state_len": len(trainer.q_optimizer.state),
            "params_equal_guide": all(torch.equal(a, b) for a, b in zip(trainer.actor.parameters(), guide.actor.parameters())),
            "target_equals_guide_qf": all(torch.equal(a, b) for a, b in zip(trainer.q_target.parameters(), guide.qf.parameters())),
            "all_curriculum_stages": [float(x) for x in cfg2.all_curriculum_stages],
            "agent_type_stage": float(cfg2.agent_type_stage)}
    one = synth.synth_transitions(1, S, A, seed=4100, antmaze_rewards=True)
    buf = ref.ReplayBuffer(S, A, 10_000, "cpu")
    buf.add_transition(one["observations"][0], one["actions"][0], float(one["rewards"][0]), one["next_observations"][0],
                       False)
    np.random.seed(77)
    batch = buf.sample(cfg.batch_size)
    assert all(torch.equal(b[0], b[-1]) for b in batch)      # 256 copies of the one row
    log = trainer.train(batch)
    out = grab(trainer, stride, True)
    out["guide_losses"] = np.array(g_losses, dtype=np.float64)
    out["losses"] = np.array([log["value_loss"], log["q_loss"], log["actor_loss"]], dtype=np.float64)
    desc["total_it_after_step"] = int(trainer.total_it)
    meta = {"kind": "jsrl_handoff", "S": S, "A": A, "gaussian": True, "B": cfg.batch_size, "seed": seed, "stride": stride,
            "hyper": hyper, "lrs": lrs, "offline_iterations": cfg.offline_iterations, "guide_batch_seed0": 4000,
            "row_seed": 4100, "np_seed": 77, "buffer_size": 10_000, **desc}
    out["meta"] = np.array(json.dumps(meta))
    np.savez(os.path.join(outdir, name + ".npz"), **out)
    print(f"{name}: handoff losses={out['losses']}")


def jsrl_hostlogic_case(jsrl, outdir):
    """SURVEY §8c "G9 JSRL host logic" (stored as g13: the g9_* names hold the dropout fixtures): prepare_finetuning's
    curricula, a horizon_update_callback trace over a fixed evaluation sequence and timestep_horizon's truth table —
    deterministic host logic of jsrl_utils.py:50-96, 137-174, 395-426 that must not change when `from iql import ...`
    resolves to another module."""
    import contextlib
    import io

    class Cfg(_JsrlCfg):
        n_curriculum_stages = 5

    out = {}
    cfg = jsrl.prepare_finetuning(300, Cfg())
    out["stages_time_step"] = np.asarray(cfg.all_curriculum_stages, dtype=np.float64)
    out["agent_types_disabled"] = np.asarray(cfg.all_agent_types, dtype=np.float64)
    c2 = Cfg()
    c2.no_agent_types = False
    c2.horizon_fn = "goal_dist"
    c2 = jsrl.prepare_finetuning(12.5, c2)
    out["stages_goal_dist"] = np.asarray(c2.all_curriculum_stages, dtype=np.float64)
    out["agent_types_enabled"] = np.asarray(c2.all_agent_types, dtype=np.float64)
    evals = [0.1, 0.2, 0.15, 0.3, 0.4, 0.1, 0.5, 0.6, 0.7, 0.65, 0.9, 0.95, 0.2, 0.99, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0,
             1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0]
    trace = []
    with contextlib.redirect_stdout(io.StringIO()):
        for r in evals:
            cfg = jsrl.horizon_update_callback(cfg, r)
            trace.append([cfg.curriculum_stage_idx, cfg.curriculum_stage, cfg.agent_type_stage, cfg.best_eval_score,
                          float(np.mean(cfg.rolling_mean_rews))])
    out["evals"] = np.asarray(evals, dtype=np.float64)
    out["callback_trace"] = np.asarray(trace, dtype=np.float64)
    table = []
    t = Cfg()
    t = jsrl.prepare_finetuning(300, t)
    for stage_idx in range(5):
        t.curriculum_stage_idx = stage_idx
        t.curriculum_stage = t.all_curriculum_stages[stage_idx]
        for ep_type in (0.0, 1.0, 2.0):
            t.ep_agent_type = ep_type
            for step in (0, 74, 75, 150, 299, 300, 1000):
                use, val = jsrl.timestep_horizon(step, None, None, t)
                table.append([stage_idx, ep_type, step, float(use), float(val)])
    t.curriculum_stage = np.nan
    t.ep_agent_type = 5.0
    use, val = jsrl.timestep_horizon(7, None, None, t)
    table.append([-1, 5.0, 7, float(use), float(val)])
    out["timestep_horizon_table"] = np.asarray(table, dtype=np.float64)
    out["meta"] = np.array(json.dumps({"kind": "jsrl_host_logic", "init_horizon": 300, "n_curriculum_stages": 5,
                                       "rolling_mean_n": 5, "tolerance": 0.05}))
    np.savez(os.path.join(outdir, "g13_jsrl_hostlogic.npz"), **out)
    print("g13_jsrl_hostlogic ok: final stage idx", trace[-1][0])


def bf16_largebatch_case(ref, name, outdir, stride=13):
    """configs[4]'s per-GPU share in fp32 BY THE REFERENCE (S=39, A=28, B=1024, actor dropout 0.1 with injected masks,
    iql_tau 0.8): the oracle target of the bf16-operand path at its defining size."""
    dropout_case(ref, name, 39, 28, True, 0.1, 1024, 82, stride, outdir)


def offline_surface_case(ref_root, outdir):
    """G15: the OFFLINE flavour's surface (algorithms/offline/iql.py, imported under another module name): state_dict
    keys of MLP / policies for dropout None / 0.0 / 0.1 (gate `is not None`, :287), the always-present LR schedule
    (:422, :513-537), TrainConfig's fields and defaults (:30-85), the buffer's sample bound and add_transition."""
    import dataclasses
    spec = importlib.util.spec_from_file_location("ref_offline_iql", os.path.join(ref_root, "algorithms", "offline", "iql.py"))
    off = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(off)
    desc = {"mlp_keys": {}, "policy_keys": {}}
    for tag, d in (("none", None), ("zero", 0.0), ("p10", 0.1)):
        desc["mlp_keys"][tag] = list(off.MLP([5, 7, 7, 3], dropout=d).state_dict().keys())
        desc["policy_keys"][tag] = {"gauss": list(off.GaussianPolicy(5, 3, 1.0, dropout=d).state_dict().keys()),
                                    "det": list(off.DeterministicPolicy(5, 3, 1.0, dropout=d).state_dict().keys())}
    desc["policy_default_dropout_keys"] = list(off.GaussianPolicy(5, 3, 1.0).state_dict().keys())
    q, v, a = off.TwinQ(5, 3), off.ValueFunction(5), off.GaussianPolicy(5, 3, 1.0)
    tr = off.ImplicitQLearning(1.0, a, torch.optim.Adam(a.parameters(), lr=3e-4), q, torch.optim.Adam(q.parameters(), lr=3e-4),
                               v, torch.optim.Adam(v.parameters(), lr=3e-4), max_steps=10, device="cpu")
    sd = tr.state_dict()
    desc["state_dict_keys"] = list(sd.keys())
    desc["schedule_state_keys"] = sorted(sd["actor_lr_schedule"].keys())
    desc["has_partial_load"] = hasattr(tr, "partial_load_state_dict")
    cfg = off.TrainConfig()
    desc["train_config"] = {f.name: (getattr(cfg, f.name) if f.name not in ("name", "checkpoints_path") else None)
                            for f in dataclasses.fields(cfg)}
    buf = off.ReplayBuffer(3, 2, 8, "cpu")
    try:
        buf.add_transition()
        desc["add_transition"] = "ok"
    except NotImplementedError:
        desc["add_transition"] = "NotImplementedError"
    np.savez(os.path.join(outdir, "g15_offline_surface.npz"), meta=np.array(json.dumps(desc)))
    print("g15_offline_surface ok:", desc["mlp_keys"]["zero"])


def round2_cases(ref, args):
    jsrl = import_reference_jsrl(args.ref)
    resume_case(ref, "g11_resume_S17A6_gauss", 17, 6, True, 110, 13, args.out)
    resume_case(ref, "g11_resume_S29A8_det", 29, 8, False, 112, 13, args.out)
    jsrl_handoff_case(ref, jsrl, "g12_jsrl_handoff_S29A8", 29, 8, 120, 13, args.out)
    jsrl_hostlogic_case(jsrl, args.out)
    bf16_largebatch_case(ref, "g14_c5_B1024_dropout", args.out)
    offline_surface_case(args.ref, args.out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--only", default=None, help="'act': only the G10 act fixtures; 'r2': only the round-2 fixtures "
                                                 "(g11 resume, g12 JSRL hand-off, g13 JSRL host logic, g14 config-5 batch)")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    torch.set_num_threads(1)
    ref = import_reference(args.ref)
    if args.only == "r2":
        round2_cases(ref, args)
        return

    act_case(ref, "g10_act_S17A6_gauss", 17, 6, True, 1.0, 90, args.out)
    act_case(ref, "g10_act_S29A8_det", 29, 8, False, 2.5, 91, args.out)
    act_case(ref, "g10_act_S39A28_gauss", 39, 28, True, 0.5, 92, args.out)
    if args.only == "act":
        return

    cid = 0
    for (S, A) in ((17, 6), (29, 8), (39, 28)):
        for gaussian in (True, False):
            for (beta, tq) in ((3.0, 0.7), (10.0, 0.9)):
                name = f"g1_S{S}A{A}_{'gauss' if gaussian else 'det'}_b{int(beta)}"
                single_step_case(ref, name, S, A, gaussian, beta, tq, 256, cid, 7 if cid == 0 else 13, args.out)
                cid += 1
    single_step_case(ref, "g7_edge_gauss", 17, 6, True, 3.0, 0.7, 256, 50, 13, args.out, edge=True)
    single_step_case(ref, "g7_edge_det", 17, 6, False, 10.0, 0.9, 256, 51, 13, args.out, edge=True)
    single_step_case(ref, "g1_ragged_B100", 17, 6, True, 3.0, 0.7, 100, 52, 13, args.out)
    single_step_case(ref, "g8_dp_B2048", 17, 6, True, 3.0, 0.7, 2048, 60, 13, args.out)
    dropout_case(ref, "g9_dropout_S39A28_gauss", 39, 28, True, 0.1, 256, 80, 13, args.out)
    dropout_case(ref, "g9_dropout_S17A6_det", 17, 6, False, 0.25, 256, 81, 13, args.out)
    freerun_case(ref, "g2_freerun_S17A6", 17, 6, True, 10, 256, 4096, 70, 13, args.out)
    freerun_case(ref, "g2_freerun_S29A8_det", 29, 8, False, 10, 256, 4096, 71, 13, args.out)
    gather_case(ref, args.out)
    ring_case(ref, args.out)
    lr_case(ref, args.out)
    statedict_case(ref, args.out)
    round2_cases(ref, args)


if __name__ == "__main__":
    main()
