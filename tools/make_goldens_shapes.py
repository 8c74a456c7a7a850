"""Network shapes shared by tools/make_goldens.py (reference classes) and the tests (package classes)."""


def tower_selfplay_nets(H, G, F, cfgmod):
    """h/g/f of the residual self-play fixture built from classes H/G/F and config module cfgmod
    (the reference's here, the package's in the tests)."""
    rb = lambda c: cfgmod.ResBlockConfig(out_channels=c, kernel_size=3, stride=1, padding=1, activation="relu",  # noqa: E731
                                         pool_kernel_size=1, pool_stride=1)
    d = lambda n: cfgmod.DenseLayerConfig(out_features=n, activation="relu")  # noqa: E731
    shape, A = (4, 6, 7), 7
    h = H((2, 6, 7), shape, cfgmod.RepresentationNetworkConfig(
        history_length=2,
        downsample=[cfgmod.ConvLayerConfig(out_channels=6, kernel_size=3, stride=1, padding=1, activation="relu"), rb(6)],
        res_net=[rb(5)]))
    g = G(shape, A, cfgmod.DynamicsNetworkConfig(res_net=[rb(3)], reward_net=[d(32)]))
    f = F(shape, A, cfgmod.PredictionNetworkConfig(res_net=[rb(3)], value_net=[d(32)], policy_net=[d(32)]))
    return h, g, f
