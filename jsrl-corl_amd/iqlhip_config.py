"""TrainConfig dataclasses, field-for-field with the reference so that the
pyrallis YAMLs under configs/{finetune,offline}/iql/** still parse and
`JsrlTrainConfig(TrainConfig)` (jsrl_w_iql.py:46) can subclass it unchanged.

  TrainConfig         <- algorithms/finetune/iql.py:32-69
  OfflineTrainConfig  <- algorithms/offline/iql.py:30-85
"""
from __future__ import annotations

import os
import uuid
from dataclasses import dataclass
from typing import Optional


def _unique_run_name(cfg) -> None:
    # name gets "-<env>-<8 hex>" appended, and the checkpoint dir gets the run name (iql.py:66-69)
    cfg.name = f"{cfg.name}-{cfg.env}-{str(uuid.uuid4())[:8]}"
    if cfg.checkpoints_path is not None:
        cfg.checkpoints_path = os.path.join(cfg.checkpoints_path, cfg.name)


@dataclass
class TrainConfig:
    # Experiment
    device: str = "cuda"
    env: str = "antmaze-umaze-v2"
    seed: int = 0
    eval_seed: int = 0
    eval_freq: int = int(5e4)
    n_episodes: int = 100
    offline_iterations: int = int(1e6)
    online_iterations: int = int(1e6)
    checkpoints_path: Optional[str] = None
    load_model: str = ""
    # IQL
    actor_dropout: float = 0.0
    buffer_size: int = 2_000_000
    batch_size: int = 256
    discount: float = 0.99
    tau: float = 0.005
    beta: float = 3.0
    iql_tau: float = 0.7
    expl_noise: float = 0.03
    noise_clip: float = 0.5
    iql_deterministic: bool = False
    normalize: bool = True
    normalize_reward: bool = False
    vf_lr: float = 3e-4
    qf_lr: float = 3e-4
    actor_lr: float = 3e-4
    # Wandb logging
    project: str = "jsrl-CORL-adroit"
    group: str = "IQL-D4RL"
    name: str = "IQL"

    def __post_init__(self):
        _unique_run_name(self)


@dataclass
class OfflineTrainConfig:
    project: str = "jsrl-CORL"
    group: str = "IQL-D4RL"
    name: str = "IQL"
    env: str = "halfcheetah-medium-expert-v2"
    discount: float = 0.99
    tau: float = 0.005
    beta: float = 3.0
    iql_tau: float = 0.7
    iql_deterministic: bool = False
    max_timesteps: int = int(1e6)
    buffer_size: int = 2_000_000
    batch_size: int = 256
    normalize: bool = True
    normalize_reward: bool = False
    vf_lr: float = 3e-4
    qf_lr: float = 3e-4
    actor_lr: float = 3e-4
    actor_dropout: Optional[float] = None
    eval_freq: int = int(5e3)
    n_episodes: int = 10
    checkpoints_path: Optional[str] = None
    load_model: str = ""
    seed: int = 0
    device: str = "cuda"

    def __post_init__(self):
        _unique_run_name(self)
