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


def test_position_counts_match_published_table():
    """Breadth-first enumeration with the GPU kernels themselves: the number of distinct positions
    after n plies (play stops at a four-in-a-row) equals John Tromp's published table (OEIS
    A212693) up to ply 8 -- an anchor for move legality and the win test that does not depend on
    the CPU restatement."""
    from muzero_amd.environments.connect_four import BatchedConnectFour
    want = [1, 7, 49, 238, 1120, 4263, 16422, 54859, 184275]
    root = BatchedConnectFour(1)
    root.reset()
    pieces, to_move = root.pieces.clone(), root.to_move.clone()
    counts = [1]
    for ply in range(1, len(want)):
        n = pieces.shape[0]
        env = BatchedConnectFour(n * 7)
        env.pieces.copy_(pieces.repeat_interleave(7, 0))
        env.to_move.copy_(to_move.repeat_interleave(7, 0))
        env.done.zero_()
        before = env.pieces.clone()
        act = torch.arange(7, dtype=torch.int32, device="cuda").repeat(n)
        _, done = env.step(act)
        legal = (env.pieces != before).any(dim=1)  # a full column leaves the board unchanged
        state = torch.cat([env.pieces[legal], env.to_move[legal, None].long(), done[legal, None].long()], dim=1)
        uniq = torch.unique(state, dim=0)
        counts.append(int(uniq.shape[0]))
        live = uniq[uniq[:, 3] == 0]
        pieces, to_move = live[:, :2].contiguous(), live[:, 2].to(torch.int8).contiguous()
    assert counts == want


def test_move_end_equals_pick_step_bookkeeping_and_reset():
    """mz_c4_move_end against the four calls it stands for -- mz_pick_action, mz_c4_step, mz_episode_end,
    mz_c4_reset (src/training_data_generator.py:107-163 for every game) -- over 60 moves of 600 games from the same
    draws: actions, rewards, ended flags, step counters and board states stay identical."""
    import torch
    from muzero_amd import _hip
    from muzero_amd._hip import ptr
    from muzero_amd.environments.connect_four import BatchedConnectFour
    from muzero_amd.selfplay import pick_action
    G, A, limit = 600, 7, 30
    gen = torch.Generator(device="cuda").manual_seed(9)
    a_env, b_env = BatchedConnectFour(G), BatchedConnectFour(G)
    a_env.reset(), b_env.reset()
    step_a = torch.zeros(G, dtype=torch.int32, device="cuda")
    step_b = torch.zeros(G, dtype=torch.int32, device="cuda")
    act_a = torch.zeros(G, dtype=torch.int32, device="cuda")
    act_b = torch.zeros(G, dtype=torch.int32, device="cuda")
    ends = 0
    for _ in range(60):
        policy = torch.rand(G, A, device="cuda", generator=gen)
        policy[::7, 3] = policy[::7, 5] = 2.0  # ties: the first maximum wins
        explore = (torch.rand(G, device="cuda", generator=gen) < 0.3).to(torch.int8)
        rnd = torch.randint(0, A, (G,), device="cuda", generator=gen, dtype=torch.int32)
        # the four separate calls
        pick_action(policy, explore, rnd, act_a)
        reward_a, done_a = a_env.step(act_a)
        ended_a = torch.empty(G, dtype=torch.int8, device="cuda")
        _hip.check(_hip.lib().mz_episode_end(ptr(done_a), ptr(step_a), limit, G, ptr(ended_a), None), "mz_episode_end")
        a_env.reset(ended_a)
        # the one call
        played, reward_b, ended_b = b_env.move_end(policy, explore, rnd, step_b, limit, act_b)
        torch.cuda.synchronize()
        assert torch.equal(act_a, act_b) and torch.equal(played, act_b)
        assert torch.equal(reward_a, reward_b) and torch.equal(ended_a, ended_b) and torch.equal(step_a, step_b)
        assert torch.equal(a_env.pieces, b_env.pieces) and torch.equal(a_env.to_move, b_env.to_move) and torch.equal(a_env.done, b_env.done)
        ends += int(ended_b.sum())
    assert ends > G  # games ended (wins, full columns, the step limit) and restarted along the way
