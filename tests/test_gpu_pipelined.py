"""Two-half pipelined self-play (host simulators stepped while the GPU searches the other half) against the
serial engine."""
import numpy as np
import pytest
import torch

from tests.test_host_pool import CountingEnv

pytestmark = pytest.mark.gpu


class NoisyEnv(CountingEnv):
    """CountingEnv whose frames are noise seeded by (game, step): a game's observations do not depend on which
    engine or which half plays it."""

    def __init__(self, gid, length, shape, actions, single_player=False):
        super().__init__(gid, length, shape=shape, actions=actions)
        self.single_player = single_player

    def get_to_play(self):
        return 0 if self.single_player else super().get_to_play()

    def _obs(self):
        gen = torch.Generator().manual_seed(self.gid * 1000 + self.t + 1000 * 1000 * self.resets)
        return torch.rand((1,) + self.shape, generator=gen)


def _cfg(S, eps):
    from muzero_amd.config import MCTSConfig, TrainingDataGeneratorConfig
    return TrainingDataGeneratorConfig(
        num_episodes=1, max_steps_per_episode=6, total_time=0, epsilon=eps, epsilon_decay=1.0, min_epsilon=eps,
        mcts=MCTSConfig(selection_strategy="puct", max_iterations=S, max_time=0, model_look_ahead=1,
                        discount_factor=0.998, dirichlet_alpha=0.3, noise_frac=0.25))


def _same_episodes(a, b, exact):
    assert len(a) == len(b) and len(a) > 0
    for ea, eb in zip(a, b):
        assert len(ea.chunks) == len(eb.chunks)
        for ca, cb in zip(ea.chunks, eb.chunks):
            assert torch.equal(ca.state, cb.state) and ca.best_action == cb.best_action and ca.reward == cb.reward
            if exact:
                assert torch.equal(ca.policy, cb.policy) and ca.value == cb.value
            else:
                np.testing.assert_allclose(ca.policy.numpy(), cb.policy.numpy(), rtol=1e-4, atol=1e-5)
                np.testing.assert_allclose(ca.value, cb.value, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("device_rng", [False, True], ids=["host_streams", "device_streams"])
@pytest.mark.parametrize("G", [10, 7])
def test_pipelined_selfplay_plays_the_serial_engines_games(device_rng, G):
    """MLP networks (the single-launch search computes every tree independently of its neighbours): with one seed
    the two-half engine and the serial engine record identical episodes, bit for bit -- same frames, policies,
    values, actions, rewards, episode ends (by the simulators and by max_steps_per_episode) and restarts."""
    from muzero_amd.environments.host_pool import HostEnvPool
    from muzero_amd.selfplay import PipelinedSelfPlay, SelfPlayEngine
    from tests.test_gpu_nets import _c4_nets
    h, g, f = _c4_nets(0)
    cfg = _cfg(8, 0.3)
    mk = lambda: [NoisyEnv(gi, 2 + gi % 7, (2, 6, 7), 7) for gi in range(G)]  # noqa: E731  (lengths 2..8: some hit the 6-step cap)
    serial = SelfPlayEngine(h, g, f, G, cfg, seed=5, device_rng=device_rng, env=HostEnvPool(mk(), workers=2))
    want = serial.generate(num_moves=14)
    piped = PipelinedSelfPlay(h, g, f, mk(), cfg, seed=5, device_rng=device_rng, workers=2)
    got = piped.generate(num_moves=14)
    _same_episodes(got, want, exact=True)
    assert {len(e.chunks) for e in got} <= set(range(2, 7)) and max(len(e.chunks) for e in got) == 6
    serial.env.close()
    piped.close()


def test_pipelined_selfplay_with_conv_h_and_towers():
    """Configs 3-4 in small (convolutional h with per-game BatchNorm statistics, residual-tower g / f in the
    lock-step search, single-player host simulators): the two-half engine plays the serial engine's games; network
    outputs may differ in the last bits where a kernel's row grouping depends on the batch size."""
    from muzero_amd.config import ConvLayerConfig, RepresentationNetworkConfig, ResBlockConfig
    from muzero_amd.environments.host_pool import HostEnvPool
    from muzero_amd.neural_networks import RepresentationNetwork
    from muzero_amd.selfplay import PipelinedSelfPlay, SelfPlayEngine
    from muzero_amd.towers import TowerNetworks
    from tests.test_gpu_towers import load_tower_nets
    g, f, A, shape = load_tower_nets(0)          # latent (16, 4, 4), 5 actions
    torch.manual_seed(4)
    rb = ResBlockConfig(out_channels=6, kernel_size=3, stride=1, padding=1, activation="relu", pool_kernel_size=1,
                        pool_stride=1)
    h = RepresentationNetwork((2, 8, 8), shape, RepresentationNetworkConfig(history_length=2, downsample=[
        ConvLayerConfig(out_channels=6, kernel_size=2, stride=2, padding=0, activation="relu"), rb], res_net=[]))
    G = 8
    cfg = _cfg(12, 0.0)
    mk = lambda: [NoisyEnv(gi, 3 + gi % 3, (2, 8, 8), A, single_player=True) for gi in range(G)]  # noqa: E731
    serial = SelfPlayEngine(h, g, f, G, cfg, seed=3, env=HostEnvPool(mk(), workers=2))
    assert isinstance(serial.nets, TowerNetworks)
    want = serial.generate(num_moves=9)
    piped = PipelinedSelfPlay(h, g, f, mk(), cfg, seed=3, workers=2)
    got = piped.generate(num_moves=9)
    _same_episodes(got, want, exact=False)
    serial.env.close()
    piped.close()
