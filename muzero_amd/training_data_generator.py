"""Single-environment self-play with the reference's interface
(src/training_data_generator.py:28-258): Chunk / Episode records, TrainingDataGenerator and
the pickle helpers.  The search runs on the GPU engine through MCTS.run; the environment is
any object implementing muzero_amd.environment.Environment (host simulators included).

The global RNGs are consumed in the reference's order: per move the S+1 child choices and the
Dirichlet sample inside MCTS.run, then one `random.random()` for the epsilon test and, when
exploring, one `numpy.random.choice`.
"""
from __future__ import annotations

import os
import pickle
import random
import time
from dataclasses import dataclass
from typing import cast

import numpy
import torch
from torch import Tensor

from muzero_amd.config import TrainingDataGeneratorConfig
from muzero_amd.ring_buffer import Frame, FrameRingBuffer, make_history_tensor
from muzero_amd.search.factory import create_mcts
from muzero_amd.search.nodes import Node

random.seed(0)  # the reference seeds Python's global RNG when this module is imported (:25)


@dataclass
class Chunk:
    state: Tensor       # raw observation fed to the history
    policy: Tensor      # MCTS policy (softmax of root children's value sums), float32 [A]
    reward: float       # environment reward after `best_action`
    value: float        # MCTS root mean value
    best_action: int


@dataclass
class Episode:
    chunks: list[Chunk]


class TrainingDataGenerator:
    def __init__(self, env, repr_net, dyn_net, pred_net, config: TrainingDataGeneratorConfig, device="cpu"):
        from muzero_amd.selfplay import RepresentationRunner
        self.env = env
        self.repr_net, self.dyn_net, self.pred_net = repr_net, dyn_net, pred_net
        self.device = device
        self.num_episodes = config.num_episodes
        self.max_steps = config.max_steps_per_episode
        self.mcts = create_mcts(dynamics_network=dyn_net, prediction_network=pred_net,
                                actions=_make_actions_tensor(env, device), config=config.mcts, device=device)
        self.epsilon = config.epsilon
        self.epsilon_decay = config.epsilon_decay
        self.min_epsilon = config.min_epsilon
        self.config = config
        self._h = RepresentationRunner(repr_net)

    def generate_training_data(self, training_steps: int = 0) -> list[Episode]:
        history: list[Episode] = []
        for _ in range(self.num_episodes):
            self.env.reset()
            episode = Episode(chunks=[])
            ring = FrameRingBuffer(size=self.repr_net.history_length)
            for _ in range(self.max_steps):
                state = self.env.get_state()
                frame = Frame(state=state, action=0)
                if not ring.full():
                    ring.fill(frame)
                else:
                    ring.add(frame)
                latent = self._h(make_history_tensor(ring))
                root = Node(latent_state=latent, to_play=self.env.get_to_play())
                tree_policy, value = self.mcts.run(root)
                policy_tensor = Tensor(tree_policy).to(self.device)
                best_action = self._select_action(policy_tensor, training_steps)
                _, reward, done = self.env.step(best_action)
                episode.chunks.append(Chunk(state=state, value=value, policy=policy_tensor,
                                            best_action=best_action, reward=reward))
                if done:
                    break
            history.append(episode)
        return history

    def _select_action(self, policy: Tensor, training_steps: int) -> int:
        if random.random() < self._calculate_epsilon(training_steps):
            return int(numpy.random.choice(policy.shape[0], p=self.config.random_action_probability))
        return int(policy.argmax().item())

    def _calculate_epsilon(self, training_steps: int) -> float:
        return max(self.epsilon * (self.epsilon_decay ** training_steps), self.min_epsilon)


def _make_actions_tensor(env, device) -> Tensor:
    space = env.get_action_space()
    if isinstance(space, int):
        return Tensor(range(space)).long().to(device)
    return Tensor(space).long().to(device)


DATA_FOLDER = "data/"


def save_training_data(training_data: list[Episode]) -> str:
    os.makedirs(DATA_FOLDER, exist_ok=True)
    filename = f"{DATA_FOLDER}training_episodes_{len(training_data)}_at_{time.time()}.pkl"
    with open(filename, "wb") as fh:
        pickle.dump(training_data, fh)
    return filename


def load_training_data(path: str) -> list[Episode]:
    """Unpickles a file written by save_training_data (only load files you wrote)."""
    with open(path, "rb") as fh:
        episodes = pickle.load(fh)
    _validate_data(episodes)
    return cast(list[Episode], episodes)


def load_all_training_data() -> list[Episode]:
    episodes: list[Episode] = []
    for name in os.listdir(DATA_FOLDER):
        path = os.path.join(DATA_FOLDER, name)
        if os.path.getsize(path) == 0:
            print(f"Skipping empty file: {path}")
            continue
        with open(path, "rb") as fh:
            try:
                episodes.extend(pickle.load(fh))
            except EOFError:
                print(f"Error loading {path}: file is corrupted or incomplete.")
    _validate_data(episodes)
    return episodes


def _validate_data(data) -> None:
    if not isinstance(data, list):
        raise ValueError(f"Expected a list of Episode objects, but got {type(data)}")
    for ep in data:
        if not isinstance(ep, Episode):
            raise ValueError(f"Expected an Episode object, but got {type(ep)}")


def delete_all_training_data() -> None:
    names = os.listdir(DATA_FOLDER)
    print(f"Deleting {len(names)} training data files...")
    for name in names:
        os.remove(os.path.join(DATA_FOLDER, name))
    print("All training data files deleted.")
