"""h / g / f networks with the reference's constructor, forward and state_dict contract.

RepresentationNetwork (h): src/neural_networks/neural_network.py:28-102
DynamicsNetwork (g):       src/neural_networks/neural_network.py:105-188
PredictionNetwork (f):     src/neural_networks/neural_network.py:191-262

Sub-module attribute names (and construction order, so that a given torch seed yields
the same initial weights) match the reference, which makes `.pth` state_dicts
interchangeable.  One deliberate difference: the reference appends the output layer to
the caller's config list in place (neural_network.py:149,226,231), stacking extra layers
when one config object builds two networks; here the list is copied first.

`forward` is the stock-PyTorch path (autograd, CPU or GPU) used by the learner.  The
search engine evaluates MLP-only networks (`res_net: []`) with the fused HIP kernels via
muzero_amd.engine.PackedNetworks.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from muzero_amd.config import (
    DenseLayerConfig,
    DynamicsNetworkConfig,
    PredictionNetworkConfig,
    RepresentationNetworkConfig,
)
from muzero_amd.neural_networks.network_builder import ResBlock, build_downsample_layer, build_mlp


def _tower(res_cfgs, in_channels: int) -> tuple[nn.ModuleList, int]:
    blocks = nn.ModuleList()
    for cfg in res_cfgs:
        blocks.append(ResBlock(cfg, in_channels=in_channels))
        in_channels = cfg.out_channels
    return blocks, in_channels


class RepresentationNetwork(nn.Module):
    def __init__(self, observation_space, latent_shape, config: RepresentationNetworkConfig):
        super().__init__()
        self.history_length = config.history_length
        k = config.history_length
        channels = observation_space[0] * k + k  # K frames + K action planes
        self.downsample_layers = nn.ModuleList()
        for layer_cfg in config.downsample:
            layer, channels = build_downsample_layer(layer_cfg, channels)
            self.downsample_layers.append(layer)
        self.res_blocks, channels = _tower(config.res_net, channels)
        self.final_conv = nn.Conv2d(channels, latent_shape[0], 1) if channels != latent_shape[0] else nn.Identity()
        self.latent_height = latent_shape[1]
        self.latent_width = latent_shape[2]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        for layer in self.downsample_layers:
            x = layer(x)
        for block in self.res_blocks:
            x = block(x)
        x = self.final_conv(x)
        return F.adaptive_avg_pool2d(x, (self.latent_height, self.latent_width))


class DynamicsNetwork(nn.Module):
    def __init__(self, latent_shape, num_actions: int, config: DynamicsNetworkConfig):
        super().__init__()
        self.latent_shape = latent_shape
        self.num_actions = num_actions
        c, h, w = latent_shape
        d = c * h * w
        self.action_embedding = nn.Embedding(num_actions, d)
        self.fc_merge = nn.Linear(2 * d, d)
        self.res_blocks, channels = _tower(config.res_net, c)
        self.res_final = nn.Conv2d(channels, c, kernel_size=1) if channels != c else nn.Identity()
        head = list(config.reward_net) + [DenseLayerConfig(out_features=1, activation="linear")]
        self.reward_mlp, _ = build_mlp(head, input_dim=d)

    def forward(self, latent_state: torch.Tensor, action: torch.Tensor):
        b, c, h, w = latent_state.shape
        merged = torch.cat([latent_state.view(b, -1), self.action_embedding(action)], dim=1)
        x = F.relu(self.fc_merge(merged)).view(b, c, h, w)
        for block in self.res_blocks:
            x = block(x)
        next_latent = self.res_final(x)
        return next_latent, self.reward_mlp(next_latent.view(b, -1))


class PredictionNetwork(nn.Module):
    def __init__(self, latent_shape, num_actions: int, config: PredictionNetworkConfig):
        super().__init__()
        self.latent_shape = latent_shape
        self.num_actions = num_actions
        c, h, w = latent_shape
        d = c * h * w
        self.res_blocks, channels = _tower(config.res_net, c)
        self.res_final = nn.Conv2d(channels, c, kernel_size=1) if channels != c else nn.Identity()
        v_head = list(config.value_net) + [DenseLayerConfig(out_features=1, activation="linear")]
        self.value_mlp, _ = build_mlp(v_head, input_dim=d)
        p_head = list(config.policy_net) + [DenseLayerConfig(out_features=num_actions, activation="softmax")]
        self.policy_mlp, _ = build_mlp(p_head, input_dim=d)

    def forward(self, latent_state: torch.Tensor):
        b, c, h, w = latent_state.shape
        assert (c, h, w) == tuple(self.latent_shape), "Latent shape mismatch."
        x = latent_state
        for block in self.res_blocks:
            x = block(x)
        flat = self.res_final(x).view(b, -1)
        value = self.value_mlp(flat)
        policy = self.policy_mlp(flat)  # already softmaxed (last activation is Softmax)
        return policy, value


def load_networks(model_folder_path: str, config=None, observation_space=None, num_actions=None):
    """Rebuild h/g/f from a config and load `repr.pth`/`dyn.pth`/`pred.pth`
    (src/neural_networks/neural_network.py:265-295).  Unlike the reference this does not
    create an environment to learn the observation shape; pass it, or a Connect-Four
    config is assumed."""
    from muzero_amd.config import load_config

    config = config or load_config("config_c4_mlp.yaml")
    observation_space = observation_space or (2, 6, 7)
    num_actions = num_actions or 7
    nets = (
        RepresentationNetwork(observation_space, config.networks.latent_shape, config.networks.representation),
        DynamicsNetwork(config.networks.latent_shape, num_actions, config.networks.dynamics),
        PredictionNetwork(config.networks.latent_shape, num_actions, config.networks.prediction),
    )
    for net, name in zip(nets, ("repr", "dyn", "pred")):
        net.load_state_dict(torch.load(f"{model_folder_path}/{name}.pth", weights_only=True))
    return nets
