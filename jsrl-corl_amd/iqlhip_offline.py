"""The offline flavour of the IQL classes (reference: algorithms/offline/iql.py), where it differs from the finetune
flavour the rest of this package mirrors:

  MLP / policies   dropout is applied when `dropout is not None` (offline :287; finetune: `> 0.0`, :332) and defaults
                   to None — so `MLP(dropout=0.0)` DOES get nn.Dropout(0.0) layers and state_dict keys
                   net.{0,3,6}.* (a checkpoint written by the offline script loads only into this flavour).
  ImplicitQLearning  the cosine schedule is built unconditionally (:422; `max_steps=None` is an error exactly as in
                   the reference) and always part of state_dict / load_state_dict (:513-537); no
                   partial_load_state_dict.
  ReplayBuffer     sample() bounded by min(size, pointer), add_transition() raises NotImplementedError (:172-184).
  eval_actor       returns the array of episode returns only (:212-228).
The gradient step itself is the same arithmetic (offline :434-511 == finetune :482-563) and runs in libiqlhip.so.
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import numpy as np
import torch.nn as nn
from torch.optim.lr_scheduler import CosineAnnealingLR

import iqlhip_hostutil as _host
import iqlhip_networks as _nets
import iqlhip_trainer as _tr


class MLP(_nets.MLP):
    def __init__(self, dims, activation_fn=nn.ReLU, output_activation_fn=None, squeeze_output: bool = False,
                 dropout: Optional[float] = None):
        super().__init__(dims, activation_fn, output_activation_fn, squeeze_output, dropout)

    @staticmethod
    def _wants_dropout(dropout) -> bool:
        return dropout is not None          # offline/iql.py:287


class GaussianPolicy(_nets.GaussianPolicy):
    _MLP = MLP

    def __init__(self, state_dim: int, act_dim: int, max_action: float, hidden_dim: int = 256, n_hidden: int = 2,
                 dropout: Optional[float] = None):
        super().__init__(state_dim, act_dim, max_action, hidden_dim, n_hidden, dropout)


class DeterministicPolicy(_nets.DeterministicPolicy):
    _MLP = MLP

    def __init__(self, state_dim: int, act_dim: int, max_action: float, hidden_dim: int = 256, n_hidden: int = 2,
                 dropout: Optional[float] = None):
        super().__init__(state_dim, act_dim, max_action, hidden_dim, n_hidden, dropout)


class ImplicitQLearning(_tr.ImplicitQLearning):
    def __init__(self, max_action, actor, actor_optimizer, q_network, q_optimizer, v_network, v_optimizer,
                 iql_tau: float = 0.7, beta: float = 3.0, max_steps: int = 1000000, discount: float = 0.99,
                 tau: float = 0.005, device: str = "cpu"):
        super().__init__(max_action, actor, actor_optimizer, q_network, q_optimizer, v_network, v_optimizer,
                         iql_tau=iql_tau, beta=beta, max_steps=max_steps, discount=discount, tau=tau, device=device)
        if self.actor_lr_schedule is None:
            # offline/iql.py:422 builds the schedule unconditionally; with max_steps=None torch accepts the
            # constructor and fails at the first scheduler step — so does this class (at its first train()).
            self.actor_lr_schedule = CosineAnnealingLR(self.actor_optimizer, max_steps)

    def state_dict(self) -> Dict[str, Any]:
        sd = super().state_dict()
        sd["actor_lr_schedule"] = self.actor_lr_schedule.state_dict()
        return sd

    partial_load_state_dict = None          # not part of the offline class (offline/iql.py:513-537)


def eval_actor(env, actor: nn.Module, device: str, n_episodes: int, seed: int) -> np.ndarray:
    returns, _ = _host.eval_actor(env, actor, device, n_episodes, seed)
    return returns
