"""ORACLE — test infrastructure, NOT product code.

CPU restatement (closed-form numpy, no autograd) of the reference's IQL
gradient step.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product path (jsrl-corl_amd/)
never does and fails loudly when the HIP library is missing.

Parity pin: the reference ships no tests or golden vectors for this path
(SURVEY.md §4), so this restatement is pinned against outputs of the reference
itself, captured by tools/make_goldens.py (which imports
/root/reference/algorithms/finetune/iql.py in the build container) and
committed under tests/golden/.  tests/test_oracle_golden.py is that check.

What each function follows (paths relative to /root/reference):
  mlp_forward            algorithms/finetune/iql.py:314-344  (MLP: Linear-ReLU-Linear-ReLU-Linear)
  q_input                algorithms/finetune/iql.py:425-429  (TwinQ.both: cat([s,a],1))
  iql_losses_and_grads   algorithms/finetune/iql.py:482-563  (_update_v/_update_q/_update_policy/train)
                         :301-302 asymmetric_l2_loss, :366-369 GaussianPolicy.forward,
                         torch.distributions.Normal.log_prob (third-party torch)
  adam_update            torch/optim/adam.py _single_tensor_adam (third-party torch; the
                         optimisers are built at algorithms/finetune/iql.py:673-675 with lr only)
  polyak                 algorithms/finetune/iql.py:72-74   (soft_update, uses POST-step qf)
  cosine_lr              torch CosineAnnealingLR (closed form; reference builds it at iql.py:471)
  replay_sample          algorithms/finetune/iql.py:171-178 (np.random.randint + 5 gathers)

The three updates of one step are mutually independent (each loss touches a
disjoint parameter set and all forwards use PRE-step parameters), so the step
is restated as: all forwards -> all gradients -> three Adam updates -> Polyak.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np

EXP_ADV_MAX = 100.0      # algorithms/finetune/iql.py:26
LOG_STD_MIN = -20.0      # :27
LOG_STD_MAX = 2.0        # :28
NETS = ("vf", "q1", "q2", "pi")
GROUP_OF = {"vf": "v", "q1": "q", "q2": "q", "pi": "pi"}


def mlp_forward(p: Dict[str, np.ndarray], x: np.ndarray, masks=None):
    """Returns (out[B,d_out], h0[B,H], h1[B,H]); weights are [out,in].

    masks = (m0, m1): dropout multipliers of the two hidden activations (0 or 1/(1-p), as
    torch.nn.Dropout applies them after each ReLU of the actor, iql.py:331-333); h0/h1 are returned
    POST-dropout, which is what the next layer and the backward pass consume."""
    h0 = np.maximum(x @ p["w0"].T + p["b0"], 0)
    if masks is not None:
        h0 = h0 * masks[0].astype(h0.dtype)
    h1 = np.maximum(h0 @ p["w1"].T + p["b1"], 0)
    if masks is not None:
        h1 = h1 * masks[1].astype(h1.dtype)
    out = h1 @ p["w2"].T + p["b2"]
    return out, h0, h1


def mlp_backward(p: Dict[str, np.ndarray], x, h0, h1, dout, masks=None):
    """dout[B,d_out] -> grads dict for w0,b0,w1,b1,w2,b2 (ReLU mask = activation > 0).
    With dropout, h0/h1 are post-dropout and the chain rule carries the same multipliers."""
    g = {}
    m0 = 1 if masks is None else masks[0].astype(h0.dtype)
    m1 = 1 if masks is None else masks[1].astype(h1.dtype)
    g["w2"] = dout.T @ h1
    g["b2"] = dout.sum(0)
    # pre-activation sign: h1_post = relu(pre)*m1, so (pre > 0) == (h1_post > 0) wherever m1 != 0
    dh1 = (dout @ p["w2"]) * m1 * (h1 > 0)
    g["w1"] = dh1.T @ h0
    g["b1"] = dh1.sum(0)
    dh0 = (dh1 @ p["w1"]) * m0 * (h0 > 0)
    g["w0"] = dh0.T @ x
    g["b0"] = dh0.sum(0)
    return g


def q_input(s, a):
    return np.concatenate([s, a], axis=1)


def iql_losses_and_grads(params, batch, hyper, dtype=np.float32, grad_scale_rows: Optional[int] = None,
                         actor_masks=None):
    """All forwards + closed-form gradients from PRE-step parameters.

    batch: dict s[B,S] a[B,A] r[B] ns[B,S] d[B].  hyper: iql_tau, beta, discount,
    deterministic(bool).  `grad_scale_rows` = the divisor of the batch means
    (defaults to B; a data-parallel shard passes the GLOBAL batch size so that
    summing shard gradients reproduces the big-batch gradient — SURVEY §8e).
    """
    f = dtype
    P = {n: {k: v.astype(f) for k, v in t.items()} for n, t in params.items()}
    s, a, r, ns, d = (batch[k].astype(f) for k in ("s", "a", "r", "ns", "d"))
    B = s.shape[0]
    Bdiv = f(grad_scale_rows if grad_scale_rows is not None else B)
    tau_q = f(hyper["iql_tau"])
    beta = f(hyper["beta"])
    gamma = f(hyper["discount"])

    # forwards (iql.py:552-553, 484-487, 507, 525)
    nv, _, _ = mlp_forward(P["vf"], ns)
    nv = nv[:, 0]
    v, v_h0, v_h1 = mlp_forward(P["vf"], s)
    v = v[:, 0]
    sa = q_input(s, a)
    qt1 = mlp_forward(P["qt1"], sa)[0][:, 0]
    qt2 = mlp_forward(P["qt2"], sa)[0][:, 0]
    tq = np.minimum(qt1, qt2)
    q1, q1_h0, q1_h1 = mlp_forward(P["q1"], sa)
    q2, q2_h0, q2_h1 = mlp_forward(P["q2"], sa)
    q1 = q1[:, 0]
    q2 = q2[:, 0]
    pre, pi_h0, pi_h1 = mlp_forward(P["pi"], s, masks=actor_masks)
    mu = np.tanh(pre)

    # value loss (iql.py:489-490, 301-302)
    adv = tq - v
    wgt = np.abs(tau_q - (adv < 0).astype(f))
    v_loss = np.sum(wgt * adv * adv, dtype=f) / f(B)
    dv = (f(-2.0) * wgt * adv) / Bdiv

    # q loss (iql.py:506-508)
    y = r + ((f(1.0) - d) * gamma) * nv
    e1 = q1 - y
    e2 = q2 - y
    q_loss = (np.sum(e1 * e1, dtype=f) / f(B) + np.sum(e2 * e2, dtype=f) / f(B)) / f(2.0)
    dq1 = e1 / Bdiv
    dq2 = e2 / Bdiv

    # policy loss (iql.py:524-534)
    with np.errstate(over="ignore"):
        w = np.minimum(np.exp(beta * adv), f(EXP_ADV_MAX))
    diff = a - mu
    g_pi_extra = {}
    if hyper.get("deterministic", False):
        bc = np.sum(diff * diff, axis=1, dtype=f)
        dmu = (f(-2.0) * w[:, None] * diff) / Bdiv
    else:
        ls_raw = P["pi"]["log_std"]
        ls = np.clip(ls_raw, f(LOG_STD_MIN), f(LOG_STD_MAX))
        sig = np.exp(ls)
        var = sig * sig
        # -Normal.log_prob = (a-mu)^2/(2 var) + log(sigma) + log(sqrt(2 pi))
        nlp = diff * diff / (f(2.0) * var) + ls + f(math.log(math.sqrt(2.0 * math.pi)))
        bc = np.sum(nlp, axis=1, dtype=f)
        dmu = (-(w[:, None]) * diff / var) / Bdiv
        inside = ((ls_raw >= f(LOG_STD_MIN)) & (ls_raw <= f(LOG_STD_MAX))).astype(f)
        dls = np.sum(w[:, None] * (f(1.0) - diff * diff / var), axis=0, dtype=f) / Bdiv
        g_pi_extra["log_std"] = dls * inside
    pi_loss = np.sum(w * bc, dtype=f) / f(B)
    dpre = dmu * (f(1.0) - mu * mu)

    grads = {
        "vf": mlp_backward(P["vf"], s, v_h0, v_h1, dv[:, None]),
        "q1": mlp_backward(P["q1"], sa, q1_h0, q1_h1, dq1[:, None]),
        "q2": mlp_backward(P["q2"], sa, q2_h0, q2_h1, dq2[:, None]),
        "pi": mlp_backward(P["pi"], s, pi_h0, pi_h1, dpre, masks=actor_masks),
    }
    grads["pi"].update(g_pi_extra)
    return {
        "value_loss": v_loss, "q_loss": q_loss, "actor_loss": pi_loss,
        "next_v": nv, "target_q": tq, "v": v, "adv": adv, "q1": q1, "q2": q2,
        "mu": mu, "exp_adv": w, "grads": grads,
        "acts": {"vf": (v_h0, v_h1), "q1": (q1_h0, q1_h1), "q2": (q2_h0, q2_h1), "pi": (pi_h0, pi_h1)},
    }


def adam_scalars(lr: float, t: int, beta1: float = 0.9, beta2: float = 0.999):
    """Host-side float64 scalars of torch's _single_tensor_adam for step count t (>=1)."""
    bc1 = 1.0 - beta1 ** t
    bc2 = 1.0 - beta2 ** t
    return lr / bc1, bc2 ** 0.5


def adam_update(p, g, m, v, lr: float, t: int, beta1=0.9, beta2=0.999, eps=1e-8, dtype=np.float32):
    """One torch.optim.Adam step (defaults: no weight decay / amsgrad); returns (p,m,v)."""
    f = dtype
    step_size, bc2_sqrt = adam_scalars(lr, t, beta1, beta2)
    m = m + f(1.0 - beta1) * (g - m)                      # exp_avg.lerp_(grad, 1-beta1)
    v = v * f(beta2) + (f(1.0 - beta2) * g) * g           # mul_(beta2).addcmul_(g,g,1-beta2)
    denom = np.sqrt(v) / f(bc2_sqrt) + f(eps)
    p = p + f(-step_size) * (m / denom)                   # addcdiv_(m, denom, value=-step_size)
    return p.astype(f), m.astype(f), v.astype(f)


def polyak(target, source, tau: float, dtype=np.float32):
    f = dtype
    return (f(1.0 - tau) * target + f(tau) * source).astype(f)


def cosine_lr(base_lr: float, t: int, T: int) -> float:
    """Closed form of CosineAnnealingLR(eta_min=0) after t scheduler steps."""
    return base_lr * (1.0 + math.cos(math.pi * t / T)) / 2.0


def new_opt_state(params):
    return {
        "m": {n: {k: np.zeros_like(v) for k, v in params[n].items()} for n in NETS},
        "v": {n: {k: np.zeros_like(v) for k, v in params[n].items()} for n in NETS},
        "t": {"v": 0, "q": 0, "pi": 0},
    }


def iql_step(params, opt, batch, hyper, lrs, dtype=np.float32, grads_override=None, actor_masks=None):
    """One full reference step.  Mutates nothing; returns (new_params, new_opt, info).

    lrs: {"v","q","pi"} learning rates USED by this step (the cosine schedule
    is stepped by the caller AFTER the actor update, iql.py:539-540).
    """
    f = dtype
    info = iql_losses_and_grads(params, batch, hyper, dtype=dtype, actor_masks=actor_masks)
    grads = grads_override if grads_override is not None else info["grads"]
    newp = {n: {k: v.astype(f) for k, v in t.items()} for n, t in params.items()}
    newo = {"m": {}, "v": {}, "t": dict(opt["t"])}
    for grp in ("v", "q", "pi"):
        newo["t"][grp] += 1
    for n in NETS:
        newo["m"][n] = {}
        newo["v"][n] = {}
        grp = GROUP_OF[n]
        for k in params[n]:
            p_, m_, v_ = adam_update(
                newp[n][k], grads[n][k].astype(f), opt["m"][n][k].astype(f), opt["v"][n][k].astype(f),
                lrs[grp], newo["t"][grp], hyper.get("adam_beta1", 0.9), hyper.get("adam_beta2", 0.999),
                hyper.get("adam_eps", 1e-8), dtype=dtype)
            newp[n][k], newo["m"][n][k], newo["v"][n][k] = p_, m_, v_
    for src, dst in (("q1", "qt1"), ("q2", "qt2")):
        for k in params[dst]:
            newp[dst][k] = polyak(newp[dst][k], newp[src][k], hyper["tau"], dtype=dtype)
    return newp, newo, info


def actor_act(pi: Dict[str, np.ndarray], states: np.ndarray, max_action: float, noise=None, dtype=np.float32):
    """GaussianPolicy.act / DeterministicPolicy.act restated for a batch of rows (reference
    algorithms/finetune/iql.py:371-379 and :404-413; mean = MLP with Tanh output :355-361, std :366):
        a = tanh(MLP(s));  [training-mode Gaussian: a = a + exp(clamp(log_std, -20, 2)) * noise];
        action = clamp(max_action * a, -max_action, +max_action)
    `noise` stands for the N(0,1) draw inside dist.sample() (:376).  Dropout is not modelled (eval forward)."""
    x = np.asarray(states, dtype=dtype).reshape(-1, pi["w0"].shape[1])
    pre, _, _ = mlp_forward({k: v.astype(dtype) for k, v in pi.items() if k != "log_std"}, x)
    a = np.tanh(pre)
    if noise is not None:
        std = np.exp(np.clip(pi["log_std"].astype(dtype), LOG_STD_MIN, LOG_STD_MAX))
        a = a + std * np.asarray(noise, dtype=dtype)
    return np.clip(a * dtype(max_action), -max_action, max_action).astype(dtype)


def replay_sample(data: Dict[str, np.ndarray], indices: np.ndarray):
    """ReplayBuffer.sample given the drawn indices: returns s,a,r(B,1),ns,d(B,1)."""
    return (data["observations"][indices], data["actions"][indices],
            data["rewards"][indices][:, None], data["next_observations"][indices],
            data["terminals"][indices][:, None].astype(np.float32))
