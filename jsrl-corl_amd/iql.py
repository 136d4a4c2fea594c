"""Drop-in module for the reference's `algorithms/finetune/iql.py`: put this
directory on sys.path ahead of the reference's and `from iql import ...` in
jsrl_w_iql.py / jsrl_utils.py / variance_learner.py resolves here
(INTEGRATION.md).  Every name those files import is re-exported.
"""
from typing import Any, Callable, Dict, List, Optional, Tuple, Union  # noqa: F401  (re-exported: jsrl_w_iql imports Tuple)

import torch.nn as nn  # noqa: F401  (re-exported: jsrl_w_iql imports nn)

from iqlhip_config import OfflineTrainConfig, TrainConfig  # noqa: F401
from iqlhip_hostutil import (ENVS_WITH_GOAL, asymmetric_l2_loss, compute_mean_std, eval_actor,  # noqa: F401
                             is_goal_reached, modify_reward, modify_reward_online, normalize_states,
                             return_reward_range, set_env_seed, set_seed, soft_update, wandb_init, wrap_env)
from iqlhip_networks import (LOG_STD_MAX, LOG_STD_MIN, MLP, DeterministicPolicy, GaussianPolicy, Squeeze,  # noqa: F401
                             TwinQ, ValueFunction)
from iqlhip_replay import OfflineReplayBuffer, ReplayBuffer, TensorBatch  # noqa: F401
from iqlhip_trainer import EXP_ADV_MAX, ImplicitQLearning  # noqa: F401
