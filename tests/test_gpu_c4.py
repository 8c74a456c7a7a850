"""GPU parity of the bitboard Connect-Four kernels against the CPU rule restatement:
thousands of random games (illegal moves included) stepped in lock step, comparing reward,
termination, player to move and the full observation after every move."""
import numpy as np
import pytest
import torch

from oracle.c4 import C4

pytestmark = pytest.mark.gpu


def test_random_games_match_oracle():
    from muzero_amd.environments.connect_four import BatchedConnectFour
    G, T = 512, 48
    rng = np.random.default_rng(3)
    env = BatchedConnectFour(G)
    env.reset()
    cpu = [C4() for _ in range(G)]
    # bias some games towards one column so that full-column (illegal) moves occur
    bias = rng.random(G) < 0.2
    for t in range(T):
        act = rng.integers(0, 7, size=G).astype(np.int32)
        act[bias] = (act[bias] % 2)
        reward, done = env.step(torch.from_numpy(act).cuda())
        obs, to_play = env.observe()
        reward, done, obs, to_play = reward.cpu().numpy(), done.cpu().numpy(), obs.cpu().numpy(), to_play.cpu().numpy()
        for g in range(G):
            r, d = cpu[g].step(int(act[g]))
            assert (r, d) == (float(reward[g]), bool(done[g])), (t, g)
            assert cpu[g].to_play == int(to_play[g]), (t, g)
            assert np.array_equal(cpu[g].observe(), obs[g]), (t, g)
    assert done.all()  # 48 > 42 moves: every game has ended
    # masked reset restarts only the selected games
    mask = torch.zeros(G, dtype=torch.int8, device="cuda")
    mask[::2] = 1
    env.reset(mask)
    obs, _ = env.observe()
    assert obs[::2].abs().sum().item() == 0 and (obs[1::2].sum(dim=(1, 2, 3)) > 0).all()
    assert env.done[::2].sum().item() == 0 and env.done[1::2].all()


def test_single_game_adapter_interface():
    from muzero_amd.environments.connect_four import ConnectFour
    env = ConnectFour()
    assert env.get_observation_space() == (2, 6, 7) and env.get_action_space() == tuple(range(7))
    s = env.reset()
    assert s.shape == (1, 2, 6, 7) and s.sum().item() == 0 and env.get_to_play() == 1
    for m in [0, 1, 0, 1, 0, 1]:
        s, r, d = env.step(m)
        assert (r, d) == (0.0, False)
    s, r, d = env.step(0)
    assert (r, d) == (-1.0, True) and env.get_to_play() == -1
