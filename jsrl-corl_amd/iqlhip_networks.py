"""IQL network classes with the reference's constructor signatures and
state_dict key names (reference: algorithms/finetune/iql.py:305-442), so that
checkpoints and `isinstance` checks in jsrl_utils.py:593,603 keep working.

These modules only *hold* parameters (as views into the trainer's flat arena
once an ImplicitQLearning is built).  The training-step arithmetic is NOT done
here: it runs in libiqlhip.so, and so does `act()` (the B=1 env-interaction
path, iql.py:371-379 / 404-413) once a GPU trainer owns the actor; before
that, or on a CPU device, `act()` and forward() are plain PyTorch exactly as
in the reference (environment loops iql.py:725-738, tests).
"""
from __future__ import annotations

import weakref
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
from torch.distributions import Normal

LOG_STD_MIN = -20.0
LOG_STD_MAX = 2.0


class Squeeze(nn.Module):
    """x.squeeze(dim) as a layer (iql.py:305-311)."""

    def __init__(self, dim: int = -1):
        super().__init__()
        self.dim = dim

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x.squeeze(dim=self.dim)


class MLP(nn.Module):
    """Linear -> act [-> Dropout] ... -> Linear [-> out act] [-> squeeze] (iql.py:314-344).

    `self.net` is an nn.Sequential so parameter keys are net.{0,2,4}.* without
    dropout (net.{0,3,6}.* with), exactly like the reference.
    """

    def __init__(self, dims: Sequence[int], activation_fn: Callable[[], nn.Module] = nn.ReLU,
                 output_activation_fn: Optional[Callable[[], nn.Module]] = None,
                 squeeze_output: bool = False, dropout: Optional[float] = 0.0):
        super().__init__()
        if len(dims) < 2:
            raise ValueError("MLP requires at least two dims (input and output)")
        if squeeze_output and dims[-1] != 1:
            raise ValueError("Last dim must be 1 when squeezing")
        stack: List[nn.Module] = []
        for d_in, d_out in zip(dims[:-2], dims[1:-1]):
            stack += [nn.Linear(d_in, d_out), activation_fn()]
            if self._wants_dropout(dropout):
                stack.append(nn.Dropout(dropout))
        stack.append(nn.Linear(dims[-2], dims[-1]))
        if output_activation_fn is not None:
            stack.append(output_activation_fn())
        if squeeze_output:
            stack.append(Squeeze(-1))
        self.net = nn.Sequential(*stack)

    @staticmethod
    def _wants_dropout(dropout) -> bool:
        # finetune flavour: `dropout > 0.0` (iql.py:332)
        return dropout is not None and dropout > 0.0

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.net(x)


# Policy modules whose parameters live in a libiqlhip arena -> the trainer that owns the context (set by
# ImplicitQLearning on a GPU device).  A registry instead of a module attribute: nothing is added to the
# module's __dict__/state_dict, and a dead trainer simply disappears.
_ACTOR_OWNERS: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


def register_actor_owner(actor: nn.Module, trainer) -> None:
    _ACTOR_OWNERS[actor] = weakref.ref(trainer)


def _hip_owner(actor: nn.Module, device) -> Optional[object]:
    """The trainer whose HIP context can run this actor's forward for `device`, else None (-> PyTorch path)."""
    ref = _ACTOR_OWNERS.get(actor)
    owner = ref() if ref is not None else None
    if owner is None or not owner.can_act_on(device):
        return None
    if actor.training and dropout_p(actor) > 0.0:
        return None          # training-mode dropout inside act(): the library's inference forward is eval-mode
    return owner


class _PolicyBase(nn.Module):
    _MLP = MLP          # the offline flavour (iqlhip_offline.py) substitutes its MLP (other dropout gate)

    def __init__(self, state_dim: int, act_dim: int, max_action: float, hidden_dim: int, n_hidden: int, dropout):
        super().__init__()
        self.net = self._MLP([state_dim] + [hidden_dim] * n_hidden + [act_dim], output_activation_fn=nn.Tanh,
                             dropout=dropout)
        self.max_action = max_action

    def _scaled(self, action: torch.Tensor) -> np.ndarray:
        a = torch.clamp(action * self.max_action, -self.max_action, self.max_action)
        return a.cpu().data.numpy().flatten()


class GaussianPolicy(_PolicyBase):
    """tanh-mean, state-independent log_std Gaussian actor (iql.py:347-379)."""

    def __init__(self, state_dim: int, act_dim: int, max_action: float, hidden_dim: int = 256, n_hidden: int = 2,
                 dropout: Optional[float] = 0.0):
        super().__init__(state_dim, act_dim, max_action, hidden_dim, n_hidden, dropout)
        self.log_std = nn.Parameter(torch.zeros(act_dim, dtype=torch.float32))

    def forward(self, obs: torch.Tensor) -> Normal:
        mean = self.net(obs)
        std = torch.exp(self.log_std.clamp(LOG_STD_MIN, LOG_STD_MAX))
        return Normal(mean, std)

    @torch.no_grad()
    def act(self, state: np.ndarray, device: str = "cpu"):
        owner = _hip_owner(self, device)
        if owner is not None:       # fused device path: one forward of the policy MLP + tanh/noise/scale/clamp
            return owner.act_one(state, self.max_action, sample=self.training)
        obs = torch.tensor(state.reshape(1, -1), device=device, dtype=torch.float32)
        dist = self(obs)
        action = dist.sample() if self.training else dist.mean
        return self._scaled(action)


class DeterministicPolicy(_PolicyBase):
    """tanh MLP actor (iql.py:382-413)."""

    def __init__(self, state_dim: int, act_dim: int, max_action: float, hidden_dim: int = 256, n_hidden: int = 2,
                 dropout: Optional[float] = 0.0):
        super().__init__(state_dim, act_dim, max_action, hidden_dim, n_hidden, dropout)

    def forward(self, obs: torch.Tensor) -> torch.Tensor:
        return self.net(obs)

    @torch.no_grad()
    def act(self, state: np.ndarray, device: str = "cpu"):
        owner = _hip_owner(self, device)
        if owner is not None:
            return owner.act_one(state, self.max_action, sample=False)
        obs = torch.tensor(state.reshape(1, -1), device=device, dtype=torch.float32)
        return self._scaled(self(obs))


class TwinQ(nn.Module):
    """Two independent Q MLPs over cat([s,a]) (iql.py:416-432)."""

    def __init__(self, state_dim: int, action_dim: int, hidden_dim: int = 256, n_hidden: int = 2):
        super().__init__()
        dims = [state_dim + action_dim] + [hidden_dim] * n_hidden + [1]
        self.q1 = MLP(dims, squeeze_output=True)
        self.q2 = MLP(dims, squeeze_output=True)

    def both(self, state: torch.Tensor, action: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        sa = torch.cat([state, action], 1)
        return self.q1(sa), self.q2(sa)

    def forward(self, state: torch.Tensor, action: torch.Tensor) -> torch.Tensor:
        return torch.min(*self.both(state, action))


class ValueFunction(nn.Module):
    """State-value MLP (iql.py:435-442)."""

    def __init__(self, state_dim: int, hidden_dim: int = 256, n_hidden: int = 2):
        super().__init__()
        self.v = MLP([state_dim] + [hidden_dim] * n_hidden + [1], squeeze_output=True)

    def forward(self, state: torch.Tensor) -> torch.Tensor:
        return self.v(state)


# ---------------------------------------------------------------------------
def linear_layers(module: nn.Module) -> List[nn.Linear]:
    return [m for m in module.modules() if isinstance(m, nn.Linear)]


def dropout_p(module: nn.Module) -> float:
    ps = [m.p for m in module.modules() if isinstance(m, nn.Dropout)]
    return max(ps) if ps else 0.0
