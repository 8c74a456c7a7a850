"""create_mcts (src/search/factory.py:12-57): builds the MCTS front end from an MCTSConfig."""
from __future__ import annotations

import torch

from muzero_amd.config import MCTSConfig, SelectionStrategyType
from muzero_amd.search.backpropagation import Backpropagation
from muzero_amd.search.mcts import MCTS
from muzero_amd.search.selection import PUCT, UCT
from muzero_amd.search.simulation import MuZeroSimulation


def create_mcts(dynamics_network, prediction_network, actions: torch.Tensor, config: MCTSConfig, device="cpu") -> MCTS:
    selection = UCT() if config.selection_strategy == SelectionStrategyType.uct else PUCT()
    return MCTS(
        selection=selection,
        simulation=MuZeroSimulation(dynamics_network, prediction_network, config.model_look_ahead, device),
        backpropagation=Backpropagation(config.discount_factor),
        dynamic_network=dynamics_network,
        prediction_network=prediction_network,
        actions=actions,
        max_itr=config.max_iterations,
        max_time=config.max_time,
        dirichlet_alpha=config.dirichlet_alpha,
        noise_frac=config.noise_frac,
        visualize=config.visualize,
    )
