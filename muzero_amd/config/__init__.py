"""YAML/pydantic configuration with the reference's schema.

Field names, defaults and nesting follow src/config/config_loader.py:19-159 so that the
reference's YAML files load unchanged.  The environment config models live here (the
reference keeps them beside each adapter: src/environments/connect_four.py:12-15,
car_racing.py, flappy_bird.py, lunar_lander.py) so importing the schema never imports a
simulator package.
"""
from __future__ import annotations

import enum
import os
from typing import Annotated, Literal, Optional, Union

import yaml
from pydantic import BaseModel, Field

CONFIG_DIR = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------- envs
class ConnectFourConfig(BaseModel):
    type: Literal["ConnectFour"] = "ConnectFour"
    render_mode: str = "rgb_array"
    seed: int = 42


class CarRacingConfig(BaseModel):
    type: Literal["CarRacing"] = "CarRacing"
    seed: int = 42
    render_mode: str = "rgb_array"


class FlappyBirdConfig(BaseModel):
    type: Literal["FlappyBird"] = "FlappyBird"
    seed: int = 42
    render_mode: str = "rgb_array"


class LunarLanderConfig(BaseModel):
    type: Literal["LunarLander"] = "LunarLander"
    seed: int = 42
    render_mode: str = "rgb_array"


EnvironmentConfig = Union[CarRacingConfig, ConnectFourConfig, LunarLanderConfig, FlappyBirdConfig]


# ----------------------------------------------------------------------------- search
class SelectionStrategyType(str, enum.Enum):
    uct = "uct"
    puct = "puct"

    def __str__(self) -> str:  # behaves like enum.StrEnum (py>=3.11)
        return str(self.value)


class MCTSConfig(BaseModel):
    selection_strategy: SelectionStrategyType = SelectionStrategyType.puct
    max_iterations: int
    max_time: int
    model_look_ahead: int = 5
    discount_factor: float = 1.0
    dirichlet_alpha: float = 0.3
    noise_frac: float = 0.25
    visualize: bool = False


# ----------------------------------------------------------------------------- layers
class ConvLayerConfig(BaseModel):
    type: Literal["conv_layer"] = "conv_layer"
    out_channels: int
    kernel_size: int
    stride: int
    padding: int
    activation: Optional[str] = "relu"


class PoolLayerConfig(BaseModel):
    type: Literal["pool_layer"] = "pool_layer"
    kernel_size: int
    stride: int
    pool_type: str = "max"


class ResBlockConfig(BaseModel):
    type: Literal["res_block"] = "res_block"
    out_channels: int
    kernel_size: int
    stride: int
    padding: int
    activation: str
    pool_kernel_size: int
    pool_stride: int


class DenseLayerConfig(BaseModel):
    out_features: Optional[int] = None
    activation: Optional[str] = "relu"


DownsampleLayerConfig = Annotated[
    Union[ConvLayerConfig, PoolLayerConfig, ResBlockConfig], Field(discriminator="type")
]


class RepresentationNetworkConfig(BaseModel):
    history_length: int
    downsample: list[DownsampleLayerConfig]
    res_net: list[ResBlockConfig]


class DynamicsNetworkConfig(BaseModel):
    res_net: list[ResBlockConfig]
    reward_net: list[DenseLayerConfig]


class PredictionNetworkConfig(BaseModel):
    res_net: list[ResBlockConfig]
    value_net: list[DenseLayerConfig]
    policy_net: list[DenseLayerConfig]


class NetworksConfig(BaseModel):
    latent_shape: tuple[int, int, int]
    representation: RepresentationNetworkConfig
    dynamics: DynamicsNetworkConfig
    prediction: PredictionNetworkConfig


# ----------------------------------------------------------------------------- loops
class TrainingDataGeneratorConfig(BaseModel):
    num_episodes: int
    max_steps_per_episode: int
    total_time: int
    mcts: MCTSConfig
    epsilon: float
    epsilon_decay: float
    random_action_probability: Optional[list[float]] = None
    min_epsilon: float = 0.25


class ValidationConfig(BaseModel):
    video_upload_interval: int
    simulation_count: int
    simulation_depth: int


class TrainingConfig(BaseModel):
    learning_rate: float
    weight_decay: float
    batch_size: int
    replay_buffer_size: int
    alpha: float
    discount_factor: float
    epochs: int
    betas: tuple[float, float]
    roll_ahead: int
    look_back: int
    mini_batch_size: int
    reward_coefficient: float
    value_coefficient: float
    policy_coefficient: float
    min_learning_rate: float
    total_training_steps: int
    lr_schedule: str
    optimizer: str = "sgd"
    momentum: float = 0.9
    scheduler_gamma: float = 0.99971
    scheduler_T_max: int = 40000
    scheduler_eta_min: float = 1e-5


class RunTimeConfig(BaseModel):
    use_cuda: bool


class Configuration(BaseModel):
    environment: EnvironmentConfig = Field(..., discriminator="type")
    networks: NetworksConfig
    training_data_generator: TrainingDataGeneratorConfig
    training: TrainingConfig
    validation: ValidationConfig
    runtime: RunTimeConfig
    project_name: str = "muzero"


def load_config(filename: str) -> Configuration:
    """Load `filename` from this package's config directory (or an absolute path)."""
    path = filename if os.path.isabs(filename) else os.path.join(CONFIG_DIR, filename)
    with open(path) as fh:
        return Configuration(**yaml.safe_load(fh))
