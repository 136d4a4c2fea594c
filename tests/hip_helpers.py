"""GPU-side test helpers: build the product trainer from a synthetic parameter dict,
pull tensors back as the {net:{tensor:array}} dicts the checkers use."""
from __future__ import annotations

import numpy as np
import torch

import iql  # jsrl-corl_amd/iql.py

NET_KEYS = ("w0", "b0", "w1", "b1", "w2", "b2")


def _mlp_linears(mod):
    return [m for m in mod.modules() if isinstance(m, torch.nn.Linear)]


def _load_mlp(mod, t):
    lin = _mlp_linears(mod)
    with torch.no_grad():
        for i, (w, b) in enumerate((("w0", "b0"), ("w1", "b1"), ("w2", "b2"))):
            lin[i].weight.copy_(torch.from_numpy(t[w]))
            lin[i].bias.copy_(torch.from_numpy(t[b]))


def _read_mlp(mod):
    lin = _mlp_linears(mod)
    out = {}
    for i, (w, b) in enumerate((("w0", "b0"), ("w1", "b1"), ("w2", "b2"))):
        out[w] = lin[i].weight.detach().cpu().numpy().copy()
        out[b] = lin[i].bias.detach().cpu().numpy().copy()
    return out


def build_hip_trainer(params, S, A, gaussian, hyper, lrs, max_steps, device="cuda", dropout=0.0, max_action=1.0):
    qf = iql.TwinQ(S, A)
    vf = iql.ValueFunction(S)
    actor = (iql.GaussianPolicy if gaussian else iql.DeterministicPolicy)(S, A, max_action, dropout=dropout)
    _load_mlp(vf.v, params["vf"])
    _load_mlp(qf.q1, params["q1"])
    _load_mlp(qf.q2, params["q2"])
    _load_mlp(actor.net, params["pi"])
    if gaussian:
        with torch.no_grad():
            actor.log_std.copy_(torch.from_numpy(params["pi"]["log_std"]))
    qf, vf, actor = qf.to(device), vf.to(device), actor.to(device)
    tr = iql.ImplicitQLearning(
        max_action=max_action, actor=actor, actor_optimizer=torch.optim.Adam(actor.parameters(), lr=lrs["pi"]),
        q_network=qf, q_optimizer=torch.optim.Adam(qf.parameters(), lr=lrs["q"]),
        v_network=vf, v_optimizer=torch.optim.Adam(vf.parameters(), lr=lrs["v"]),
        iql_tau=hyper["iql_tau"], beta=hyper["beta"], max_steps=max_steps, discount=hyper["discount"],
        tau=hyper["tau"], device=device)
    _load_mlp(tr.q_target.q1, params["qt1"])
    _load_mlp(tr.q_target.q2, params["qt2"])
    return tr


def read_params(tr):
    out = {"vf": _read_mlp(tr.vf.v), "q1": _read_mlp(tr.qf.q1), "q2": _read_mlp(tr.qf.q2),
           "pi": _read_mlp(tr.actor.net), "qt1": _read_mlp(tr.q_target.q1), "qt2": _read_mlp(tr.q_target.q2)}
    if hasattr(tr.actor, "log_std"):
        out["pi"]["log_std"] = tr.actor.log_std.detach().cpu().numpy().copy()
    return out


def read_moments(tr):
    tr.state_dict()  # syncs optimizer.state with the arenas
    res = {"m": {}, "v": {}}
    mods = {"vf": (tr.vf.v, tr.v_optimizer), "q1": (tr.qf.q1, tr.q_optimizer), "q2": (tr.qf.q2, tr.q_optimizer),
            "pi": (tr.actor.net, tr.actor_optimizer)}
    for net, (mod, opt) in mods.items():
        lin = _mlp_linears(mod)
        res["m"][net], res["v"][net] = {}, {}
        for i, (w, b) in enumerate((("w0", "b0"), ("w1", "b1"), ("w2", "b2"))):
            for key, p in ((w, lin[i].weight), (b, lin[i].bias)):
                st = opt.state[p]
                res["m"][net][key] = st["exp_avg"].detach().cpu().numpy().copy()
                res["v"][net][key] = st["exp_avg_sq"].detach().cpu().numpy().copy()
    if hasattr(tr.actor, "log_std"):
        st = tr.actor_optimizer.state[tr.actor.log_std]
        res["m"]["pi"]["log_std"] = st["exp_avg"].detach().cpu().numpy().copy()
        res["v"]["pi"]["log_std"] = st["exp_avg_sq"].detach().cpu().numpy().copy()
    return res


def unflatten_grads(tr, flat):
    """Flat gradient (arena order) -> {net:{tensor:array}} + the 3 loss words."""
    L = tr._layout
    names = ("vf", "q1", "q2", "pi")
    out = {}
    for i, n in enumerate(names):
        nl = L.net[i]
        k, d = nl.k_in, nl.d_out
        g = {"w1": flat[nl.w1: nl.w1 + 256 * 256].reshape(256, 256), "w0": flat[nl.w0: nl.w0 + 256 * k].reshape(256, k),
             "b0": flat[nl.b0: nl.b0 + 256], "b1": flat[nl.b1: nl.b1 + 256],
             "w2": flat[nl.w2: nl.w2 + d * 256].reshape(d, 256), "b2": flat[nl.b2: nl.b2 + d]}
        if nl.log_std >= 0:
            g["log_std"] = flat[nl.log_std: nl.log_std + d]
        out[n] = g
    return out, flat[L.n_params: L.n_params + 3]


def to_torch_batch(batch, device="cuda"):
    return [torch.from_numpy(np.ascontiguousarray(batch["s"])).to(device),
            torch.from_numpy(np.ascontiguousarray(batch["a"])).to(device),
            torch.from_numpy(np.ascontiguousarray(batch["r"][:, None])).to(device),
            torch.from_numpy(np.ascontiguousarray(batch["ns"])).to(device),
            torch.from_numpy(np.ascontiguousarray(batch["d"][:, None])).to(device)]


def head_values(tr, params, B):
    """Reconstruct next_v, v, qt1, qt2, q1, q2, pre-tanh pi from the library's head partials."""
    h = tr.debug_read("heads")
    MB, A = tr._max_batch, tr._A
    sc = h[: MB * 24].reshape(MB, 6, 4)[:B].transpose(1, 2, 0)          # [inst][ns][row]
    pi = h[MB * 24: MB * 24 + MB * A * 4].reshape(MB, A, 4)[:B].transpose(2, 0, 1)   # [ns][row][A]
    s = ((sc[:, 0] + sc[:, 1]) + sc[:, 2]) + sc[:, 3]      # the bias is folded into slice 0 by the forward kernel
    out = {"next_v": s[0], "v": s[1], "qt1": s[2], "qt2": s[3], "q1": s[4], "q2": s[5],
           "pre": ((pi[0] + pi[1]) + pi[2]) + pi[3]}
    return out
