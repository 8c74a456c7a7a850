"""Host environment pool and the per-game batch statistics of a batched h.  CPU only."""
import numpy as np
import torch

from muzero_amd.environments.host_pool import HostEnvPool


class CountingEnv:
    """Duck-typed single-game environment (the reference's Environment methods): the observation
    encodes (game id, step), an episode ends after `length` steps, reward = action."""

    def __init__(self, gid: int, length: int, shape=(2, 3, 4), actions=3):
        self.gid, self.length, self.shape, self.actions = gid, length, shape, actions
        self.t = 0
        self.resets = 0

    def _obs(self):
        return torch.full((1,) + self.shape, float(self.gid * 100 + self.t))

    def get_to_play(self):
        return 1 if self.t % 2 == 0 else -1

    def get_action_space(self):
        return (self.actions,)

    def get_observation_space(self):
        return self.shape

    def get_state(self):
        return self._obs()

    def step(self, action):
        self.t += 1
        return self._obs(), float(action), self.t >= self.length

    def reset(self):
        self.t = 0
        self.resets += 1
        return self._obs()


def test_pool_batches_observations_steps_and_restarts():
    for workers in (0, 3):
        envs = [CountingEnv(g, 2 + g) for g in range(4)]
        pool = HostEnvPool(envs, device="cpu", workers=workers)
        assert pool.num_actions == 3 and pool.observation_space == (2, 3, 4) and pool.G == 4
        obs, tp = pool.observe()
        assert obs.shape == (4, 2, 3, 4) and tp.tolist() == [1, 1, 1, 1]
        assert obs[:, 0, 0, 0].tolist() == [0.0, 100.0, 200.0, 300.0]
        r, d = pool.step(torch.tensor([0, 1, 2, 1], dtype=torch.int32))
        assert r.tolist() == [0.0, 1.0, 2.0, 1.0] and d.tolist() == [0, 0, 0, 0]
        obs2, tp2 = pool.observe()
        assert obs2[:, 0, 0, 0].tolist() == [1.0, 101.0, 201.0, 301.0] and tp2.tolist() == [-1] * 4
        assert obs[:, 0, 0, 0].tolist() == [0.0, 100.0, 200.0, 300.0]  # earlier snapshot not aliased
        r, d = pool.step(torch.tensor([2, 2, 2, 2], dtype=torch.int32))
        assert d.tolist() == [1, 0, 0, 0]
        pool.reset(d)
        obs3, _ = pool.observe()
        assert obs3[:, 0, 0, 0].tolist() == [0.0, 102.0, 202.0, 302.0]
        assert [e.resets for e in envs] == [1, 0, 0, 0]
        pool.close()


def test_batched_h_uses_per_game_statistics():
    """A batch through h inside per_call_statistics equals one call per game, which is what the
    reference does; the plain batched call does not."""
    from muzero_amd.config import ConvLayerConfig, PoolLayerConfig, RepresentationNetworkConfig, ResBlockConfig
    from muzero_amd.neural_networks import RepresentationNetwork
    from muzero_amd.selfplay import per_call_statistics
    torch.manual_seed(0)
    rb = lambda c: ResBlockConfig(out_channels=c, kernel_size=3, stride=1, padding=1, activation="relu",  # noqa: E731
                                  pool_kernel_size=1, pool_stride=1)
    cfg = RepresentationNetworkConfig(history_length=2, downsample=[
        ConvLayerConfig(out_channels=6, kernel_size=2, stride=2, padding=0, activation="relu"), rb(6), rb(8),
        PoolLayerConfig(kernel_size=2, stride=2, pool_type="max")], res_net=[rb(5)])
    h = RepresentationNetwork((3, 16, 16), (4, 2, 2), cfg)
    for m in h.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            torch.nn.init.uniform_(m.weight, 0.5, 1.5)
            torch.nn.init.uniform_(m.bias, -0.3, 0.3)
    x = torch.randn(5, 3 * 2 + 2, 16, 16)
    with torch.no_grad():
        one_by_one = torch.cat([h(x[i:i + 1]) for i in range(5)])
        running = [m.running_mean.clone() for m in h.modules() if isinstance(m, torch.nn.BatchNorm2d)]
        with per_call_statistics(h):
            batched = h(x)
        plain = h(x)
    np.testing.assert_allclose(batched.numpy(), one_by_one.numpy(), rtol=1e-5, atol=1e-6)
    assert not np.allclose(plain.numpy(), one_by_one.numpy(), rtol=1e-3, atol=1e-4)
    # the context leaves the modules as they were: forward restored, the plain call above updated the
    # running statistics again while the per-sample pass did not
    assert all("forward" not in m.__dict__ for m in h.modules())
    after = [m.running_mean for m in h.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    assert any(not torch.equal(a, b) for a, b in zip(after, running))
