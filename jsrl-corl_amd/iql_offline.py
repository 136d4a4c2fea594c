"""Drop-in module for the reference's `algorithms/offline/iql.py` (BASELINE configs[0]'s script): import it under the
name that script's callers use and every class / function it defines resolves here, in the offline
flavour (iqlhip_offline.py lists the differences from the finetune flavour of `iql.py`)."""
from typing import Any, Callable, Dict, List, Optional, Tuple, Union  # noqa: F401

import torch.nn as nn  # noqa: F401

from iqlhip_config import OfflineTrainConfig as TrainConfig  # noqa: F401
from iqlhip_hostutil import (asymmetric_l2_loss, compute_mean_std, modify_reward, normalize_states,  # noqa: F401
                             return_reward_range, set_seed, soft_update, wandb_init, wrap_env)
from iqlhip_networks import LOG_STD_MAX, LOG_STD_MIN, Squeeze, TwinQ, ValueFunction  # noqa: F401
from iqlhip_offline import MLP, DeterministicPolicy, GaussianPolicy, ImplicitQLearning, eval_actor  # noqa: F401
from iqlhip_replay import OfflineReplayBuffer as ReplayBuffer, TensorBatch  # noqa: F401
from iqlhip_trainer import EXP_ADV_MAX  # noqa: F401
