"""Layer factories for the h/g/f networks (torch definitions).

These modules define parameters, state_dict key names and the autograd path used by the
learner.  Inference inside the search engine does not run them: it reads their weights
and evaluates the networks with the HIP kernels in muzero_amd/csrc.

Behaviour follows src/neural_networks/network_builder.py:12-205: activation names
(relu is in-place), ConvLayer (conv with bias), PoolLayer (max/avg), ResBlock
(conv-BN-act-conv-BN, bias-free convs, 1x1 conv+BN skip when the shape changes,
optional max-pool) and build_mlp (Linear + activation per entry).
"""
from __future__ import annotations

import torch
from torch import nn

from muzero_amd.config import ConvLayerConfig, DenseLayerConfig, PoolLayerConfig, ResBlockConfig

_ACTIVATIONS = {
    "linear": lambda: nn.Identity(),
    "none": lambda: nn.Identity(),
    "relu": lambda: nn.ReLU(inplace=True),
    "tanh": lambda: nn.Tanh(),
    "sigmoid": lambda: nn.Sigmoid(),
    "softmax": lambda: nn.Softmax(dim=1),
}


def get_activation(activation_name: str | None) -> nn.Module:
    # The reference calls .lower() on the argument before matching, so None raises
    # AttributeError there as well (network_builder.py:17).
    key = activation_name.lower()
    if key not in _ACTIVATIONS:
        raise ValueError(f"Unknown activation function: {activation_name}")
    return _ACTIVATIONS[key]()


class ConvLayer(nn.Module):
    def __init__(self, config: ConvLayerConfig, in_channels: int):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, config.out_channels, config.kernel_size,
                              stride=config.stride, padding=config.padding)
        self.activation = get_activation(config.activation)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.activation(self.conv(x))


class PoolLayer(nn.Module):
    def __init__(self, config: PoolLayerConfig):
        super().__init__()
        kind = config.pool_type.lower()
        if kind == "max":
            self.pool = nn.MaxPool2d(kernel_size=config.kernel_size, stride=config.stride)
        elif kind == "avg":
            self.pool = nn.AvgPool2d(kernel_size=config.kernel_size, stride=config.stride)
        else:
            raise ValueError(f"Unknown pool_type: {config.pool_type}")

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.pool(x)


class ResBlock(nn.Module):
    def __init__(self, config: ResBlockConfig, in_channels: int):
        super().__init__()
        oc = config.out_channels
        self.activation = get_activation(config.activation)
        self.conv1 = nn.Conv2d(in_channels, oc, config.kernel_size, stride=config.stride,
                               padding=config.padding, bias=False)
        self.bn1 = nn.BatchNorm2d(oc)
        self.conv2 = nn.Conv2d(oc, oc, config.kernel_size, stride=1, padding=config.padding, bias=False)
        self.bn2 = nn.BatchNorm2d(oc)
        self.downsample = None
        if in_channels != oc or config.stride != 1:
            self.downsample = nn.Sequential(
                nn.Conv2d(in_channels, oc, kernel_size=1, stride=config.stride, bias=False),
                nn.BatchNorm2d(oc),
            )
        if config.pool_kernel_size > 1:
            self.pool = nn.MaxPool2d(kernel_size=config.pool_kernel_size, stride=config.pool_stride)
        else:
            self.pool = nn.Identity()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        skip = x if self.downsample is None else self.downsample(x)
        y = self.activation(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        y += skip
        return self.pool(self.activation(y))


def build_downsample_layer(layer_config, in_channels: int) -> tuple[nn.Module, int]:
    kind = layer_config.type
    if kind == "conv_layer":
        return ConvLayer(layer_config, in_channels), layer_config.out_channels
    if kind == "pool_layer":
        return PoolLayer(layer_config), in_channels
    if kind == "res_block":
        return ResBlock(layer_config, in_channels), layer_config.out_channels
    raise ValueError(f"Unknown layer type: {kind}")


def build_mlp(layers_config: list[DenseLayerConfig], input_dim: int) -> tuple[nn.Sequential, int]:
    mods: list[nn.Module] = []
    width = input_dim
    for i, cfg in enumerate(layers_config):
        if cfg.out_features is None:
            raise ValueError(f"DenseLayerConfig at index {i} has no out_features set.")
        mods += [nn.Linear(width, cfg.out_features), get_activation(cfg.activation)]
        width = cfg.out_features
    return nn.Sequential(*mods), width
