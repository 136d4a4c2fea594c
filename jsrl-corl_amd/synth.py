"""Deterministic synthetic inputs for the IQL step: D4RL-shaped transitions and
nn.Linear-distributed initial parameters, both drawn from numpy's PCG64
(`default_rng`), whose stream is stable across numpy versions.

Used by bench.py, the tests and tools/make_goldens.py so that a fixture only
has to store (seed, dims) + the reference's OUTPUTS, never the inputs.

Distributions follow SURVEY.md §8(d): obs/next_obs ~ N(0,1) (normalised states,
reference algorithms/finetune/iql.py:628-638), actions ~ U(-1,1)*0.999, rewards
~ N(0,1), dones ~ Bernoulli(p_done).  Parameter init is the distribution of
torch.nn.Linear's default init (U(+-1/sqrt(fan_in)) for W and b; reference
algorithms/finetune/iql.py:331 builds plain nn.Linear layers), log_std = 0
(algorithms/finetune/iql.py:363).
"""
from __future__ import annotations

from typing import Dict

import numpy as np

HIDDEN = 256

# Names of one MLP's tensors in arena order (see include/iqlhip.h).
MLP_TENSORS = ("w0", "b0", "w1", "b1", "w2", "b2")
NETS = ("vf", "q1", "q2", "pi")


def synth_transitions(n: int, state_dim: int, action_dim: int, seed: int = 0,
                      p_done: float = 0.01, antmaze_rewards: bool = False
                      ) -> Dict[str, np.ndarray]:
    """D4RL `qlearning_dataset`-shaped dict of float32 arrays (terminals float 0/1)."""
    rng = np.random.default_rng(seed)
    obs = rng.standard_normal((n, state_dim), dtype=np.float32)
    nobs = rng.standard_normal((n, state_dim), dtype=np.float32)
    act = (rng.random((n, action_dim), dtype=np.float32) * 2.0 - 1.0) * np.float32(0.999)
    if antmaze_rewards:
        rew = -(rng.random(n, dtype=np.float32) < 0.98).astype(np.float32)
    else:
        rew = rng.standard_normal(n, dtype=np.float32)
    done = (rng.random(n, dtype=np.float32) < p_done).astype(np.float32)
    return {
        "observations": obs,
        "actions": act.astype(np.float32),
        "rewards": rew,
        "next_observations": nobs,
        "terminals": done,
    }


def _linear(rng, fan_out: int, fan_in: int):
    bound = 1.0 / np.sqrt(fan_in)
    w = (rng.random((fan_out, fan_in), dtype=np.float32) * 2.0 - 1.0) * np.float32(bound)
    b = (rng.random((fan_out,), dtype=np.float32) * 2.0 - 1.0) * np.float32(bound)
    return w.astype(np.float32), b.astype(np.float32)


def _mlp(rng, d_in: int, d_out: int, hidden: int = HIDDEN) -> Dict[str, np.ndarray]:
    w0, b0 = _linear(rng, hidden, d_in)
    w1, b1 = _linear(rng, hidden, hidden)
    w2, b2 = _linear(rng, d_out, hidden)
    return {"w0": w0, "b0": b0, "w1": w1, "b1": b1, "w2": w2, "b2": b2}


def synth_params(state_dim: int, action_dim: int, seed: int = 0, gaussian: bool = True,
                 hidden: int = HIDDEN, perturb_target: bool = True) -> Dict[str, Dict[str, np.ndarray]]:
    """Parameter dict {vf,q1,q2,pi,qt1,qt2} -> {w0,b0,w1,b1,w2,b2[,log_std]}.

    Weights are stored [out,in] like torch.nn.Linear.  The target nets start as
    a copy of q1/q2 (reference iql.py:461) optionally perturbed a little so that
    a parity test can tell target and online nets apart.
    """
    rng = np.random.default_rng(seed + 7919)
    p = {
        "vf": _mlp(rng, state_dim, 1, hidden),
        "q1": _mlp(rng, state_dim + action_dim, 1, hidden),
        "q2": _mlp(rng, state_dim + action_dim, 1, hidden),
        "pi": _mlp(rng, state_dim, action_dim, hidden),
    }
    if gaussian:
        p["pi"]["log_std"] = np.zeros((action_dim,), dtype=np.float32)
    for src, dst in (("q1", "qt1"), ("q2", "qt2")):
        p[dst] = {}
        for k, v in p[src].items():
            if perturb_target:
                noise = rng.standard_normal(v.shape, dtype=np.float32) * np.float32(1e-2)
                p[dst][k] = (v + noise * np.abs(v)).astype(np.float32)
            else:
                p[dst][k] = v.copy()
    return p


def clone_params(p):
    return {n: {k: v.copy() for k, v in t.items()} for n, t in p.items()}


def synth_dropout_keep(n_rows: int, p: float, seed: int = 0, hidden: int = HIDDEN):
    """Two keep-masks [n_rows, hidden] (bool) for the actor's two Dropout(p) layers (tests: mask injection)."""
    rng = np.random.default_rng(seed + 104729)
    return rng.random((n_rows, hidden)) >= p, rng.random((n_rows, hidden)) >= p


def pack_keep_bits(keep: np.ndarray) -> np.ndarray:
    """bool [rows, 256] -> uint32 [rows, 8]; bit j of word w = unit 32*w + j (the layout libiqlhip uses)."""
    rows, h = keep.shape
    k = keep.reshape(rows, h // 32, 32).astype(np.uint64)
    return (k << np.arange(32, dtype=np.uint64)).sum(axis=2).astype(np.uint32)
