"""Host-side helpers the reference's loops import from `iql`
(jsrl_w_iql.py:24-41, jsrl_utils.py:16-22): dataset normalisation, reward
shaping, env wrapping, seeding.  Pure numpy / python, no device work.
gym / gymnasium / wandb are imported lazily: they are host-only and absent on
the GPU box.
"""
from __future__ import annotations

import os
import random
import uuid
from typing import Dict, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn as nn

ENVS_WITH_GOAL = ("antmaze", "pen", "door", "hammer", "relocate", "Adroit")
_LOCOMOTION = ("halfcheetah", "hopper", "walker2d")


def soft_update(target: nn.Module, source: nn.Module, tau: float):
    """Polyak averaging on modules (iql.py:72-74); the trainer does this inside the fused
    update kernel, this function exists for callers that import it."""
    with torch.no_grad():
        for t, s in zip(target.parameters(), source.parameters()):
            t.data.copy_((1 - tau) * t.data + tau * s.data)


def compute_mean_std(states: np.ndarray, eps: float) -> Tuple[np.ndarray, np.ndarray]:
    return states.mean(0), states.std(0) + eps


def normalize_states(states: np.ndarray, mean: np.ndarray, std: np.ndarray):
    return (states - mean) / std


def wrap_env(env, state_mean: Union[np.ndarray, float] = 0.0, state_std: Union[np.ndarray, float] = 1.0,
             reward_scale: float = 1.0):
    """Observation normalisation / reward scaling wrappers (iql.py:87-119), including the
    reference's handling of gymnasium's (obs, info) tuples as written there."""

    def normalize_state(state):
        info = None
        if isinstance(state, tuple):
            state = state[0]
            info = state[1]   # (sic) mirrors the reference, SURVEY Appendix A
        state = (state - state_mean) / state_std
        return state if info is None else (state, info)

    def scale_reward(reward):
        return reward_scale * reward

    if "gymnasium" in str(type(env)):
        import gymnasium
        env = gymnasium.wrappers.TransformObservation(env, normalize_state, env.observation_space)
        if reward_scale != 1.0:
            env = gymnasium.wrappers.TransformReward(env, scale_reward)
    else:
        import gym
        env = gym.wrappers.TransformObservation(env, normalize_state)
        if reward_scale != 1.0:
            env = gym.wrappers.TransformReward(env, scale_reward)
    return env


def set_env_seed(env, seed: int):
    env.seed(seed)
    env.action_space.seed(seed)


def set_seed(seed: int, env=None, deterministic_torch: bool = False):
    if env is not None:
        set_env_seed(env, seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    random.seed(seed)
    torch.manual_seed(seed)
    torch.use_deterministic_algorithms(deterministic_torch)


def wandb_init(config: dict) -> None:
    import wandb
    wandb.init(config=config, project=config["project"], group=config["group"], name=config["name"],
               id=str(uuid.uuid4()))
    wandb.run.save()


def is_goal_reached(reward: float, info: Dict) -> bool:
    for key in ("goal_achieved", "success"):
        if key in info:
            return info[key]
    return reward > 0


@torch.no_grad()
def eval_actor(env, actor: nn.Module, device: str, n_episodes: int, seed: int) -> Tuple[np.ndarray, np.ndarray]:
    env.seed(seed)
    actor.eval()
    returns, successes = [], []
    for _ in range(n_episodes):
        state, done = env.reset(), False
        total, reached = 0.0, False
        while not done:
            state, reward, done, infos = env.step(actor.act(state, device))
            total += reward
            reached = reached or is_goal_reached(reward, infos)
        successes.append(float(reached))
        returns.append(total)
    actor.train()
    return np.asarray(returns), np.mean(successes)


def return_reward_range(dataset: Dict, max_episode_steps: int) -> Tuple[float, float]:
    """min / max episode return of a D4RL dataset (semantics of iql.py:262-274): an episode ends at a terminal
    flag or after max_episode_steps steps, whichever comes first; a trailing unfinished episode is not counted.
    Episode boundaries are found per terminal-delimited run (a 1 M-row locomotion dataset has ~1 k of them), the
    returns with one segmented float64 sum — no per-transition Python loop."""
    rewards = np.asarray(dataset["rewards"], dtype=np.float64).reshape(-1)
    n = rewards.shape[0]
    term_at = np.flatnonzero(np.asarray(dataset["terminals"]).reshape(-1).astype(bool))
    run_first = np.concatenate(([0], term_at + 1))
    run_stop = np.concatenate((term_at + 1, [n]))                  # exclusive
    pieces = []
    for i, (lo, hi) in enumerate(zip(run_first.tolist(), run_stop.tolist())):
        if hi <= lo:
            continue
        cuts = np.arange(lo + max_episode_steps - 1, hi, max_episode_steps)      # time-limit ends inside the run
        if i < term_at.shape[0] and (cuts.size == 0 or cuts[-1] != hi - 1):
            cuts = np.append(cuts, hi - 1)                                        # the run's terminal step
        pieces.append(cuts)
    last = np.concatenate(pieces) if pieces else np.zeros(0, dtype=np.int64)
    if last.size == 0:
        raise ValueError("min() arg is an empty sequence")        # what the reference's min(returns) raises
    first = np.concatenate(([0], last[:-1] + 1))
    returns = np.add.reduceat(np.append(rewards, 0.0), np.append(first, last[-1] + 1))[:-1]
    return float(returns.min()), float(returns.max())


def modify_reward(dataset: Dict, env_name: str, max_episode_steps: int = 1000) -> Dict:
    if any(s in env_name for s in _LOCOMOTION):
        min_ret, max_ret = return_reward_range(dataset, max_episode_steps)
        dataset["rewards"] /= max_ret - min_ret
        dataset["rewards"] *= max_episode_steps
        return {"max_ret": max_ret, "min_ret": min_ret, "max_episode_steps": max_episode_steps}
    if "antmaze" in env_name:
        dataset["rewards"] -= 1.0
    return {}


def modify_reward_online(reward: float, env_name: str, **kwargs) -> float:
    if any(s in env_name for s in _LOCOMOTION):
        reward /= kwargs["max_ret"] - kwargs["min_ret"]
        reward *= kwargs["max_episode_steps"]
    elif "antmaze" in env_name:
        reward -= 1.0
    return reward


def asymmetric_l2_loss(u: torch.Tensor, tau: float) -> torch.Tensor:
    """Expectile loss (iql.py:301-302); the trainer computes it inside the backward kernel."""
    return torch.mean(torch.abs(tau - (u < 0).float()) * u ** 2)
