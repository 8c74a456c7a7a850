from muzero_amd.neural_networks.neural_network import (  # noqa: F401
    DynamicsNetwork,
    PredictionNetwork,
    RepresentationNetwork,
    load_networks,
)
