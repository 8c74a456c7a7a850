"""Device episode store (CPU tensors here): windows cut from the `[slot, step, ...]` tensors
equal the windows the list-based path cuts from the equivalent Episode/Chunk lists, for
history lengths 1..4 (ring-slot order of src/ring_buffer.py:11-83 via bptt's fill/add
sequence, src/training.py:252-262)."""
import numpy as np
import pytest
import torch

from muzero_amd.episode_store import DeviceEpisodeStore
from muzero_amd.learner import windows_to_batch

A, OBS = 7, (2, 6, 7)


def _z(R, V, g):
    R, V = np.asarray(R.cpu()), np.asarray(V.cpu())
    w = R.shape[1] - 1
    Z = np.zeros_like(R)
    for k in range(w + 1):
        Z[:, k] = sum(g ** (i - 1) * R[:, k + i] for i in range(w - k + 1)) + g ** (w - k) * V[:, w]
    return torch.from_numpy(Z)


def _play(store, moves, seed):
    """Random lock-step 'games' with random endings."""
    rng = np.random.default_rng(seed)
    G = store.G
    for _ in range(moves):
        obs = torch.from_numpy(rng.integers(0, 2, size=(G, 1) + OBS).astype(np.float32))
        pol = torch.from_numpy(rng.dirichlet(np.ones(A), size=G).astype(np.float32))
        val = torch.from_numpy(rng.normal(size=G))
        act = torch.from_numpy(rng.integers(0, A, size=G).astype(np.int32))
        rew = torch.from_numpy(rng.normal(size=G).astype(np.float32))
        end = torch.from_numpy((rng.random(G) < 0.06).astype(np.int8))
        store.record(obs, pol, val, act, rew, end)


@pytest.mark.parametrize("K", [1, 2, 3, 4])
def test_windows_equal_the_list_based_path(K):
    lookback, w, gamma = 4, 5, 0.998
    store = DeviceEpisodeStore(12, 30, OBS, A, capacity=64, device="cpu")
    _play(store, 80, seed=K)
    assert store.size > 10
    episodes = store.to_episodes()
    assert [len(e.chunks) for e in episodes] == store.done_len[: store.size].tolist()
    gen = torch.Generator().manual_seed(5)
    batch, slots, k = store.sample_windows(16, lookback, w, K, gamma, _z, generator=gen)
    assert ((k >= lookback) & (k + w < store.done_len[slots])).all()
    wins = [(episodes[int(s)], int(kk)) for s, kk in zip(slots, k)]
    R = np.array([[ep.chunks[i].reward for i in range(kk, kk + w + 1)] for ep, kk in wins], np.float64)
    V = np.array([[ep.chunks[i].value for i in range(kk, kk + w + 1)] for ep, kk in wins], np.float64)
    want = windows_to_batch(wins, _z(torch.from_numpy(R), torch.from_numpy(V), gamma).numpy(), R, np.ones(len(wins)), K,
                            lookback, w)
    assert torch.equal(batch.history, want.history)
    assert torch.equal(batch.actions, want.actions) and torch.equal(batch.policies, want.policies)
    torch.testing.assert_close(batch.z, want.z, rtol=0, atol=0)
    torch.testing.assert_close(batch.rewards, want.rewards, rtol=0, atol=0)


def test_ring_overwrites_oldest_and_short_episodes_are_not_eligible():
    with pytest.raises(ValueError):  # all games may end in one move: the ring must be able to take them
        DeviceEpisodeStore(4, 6, OBS, A, capacity=3, device="cpu")
    store = DeviceEpisodeStore(3, 6, OBS, A, capacity=3, device="cpu")
    _play(store, 40, seed=9)  # max_steps 6: every episode is cut at 6 moves at the latest
    assert store.size == 3 and store.n_finished > 3
    assert (store.done_len <= 6).all()
    assert store.eligible(4, 5).numel() == 0  # needs 10 chunks
    with pytest.raises(ValueError):
        store.sample_windows(2, 4, 5, 1, 0.99, _z)


def test_prioritised_sampling_follows_the_replay_buffer():
    """Probabilities, importance weights, priority updates (duplicates: the last one wins) and the
    priority of new episodes equal ReplayBuffer's (src/replay_buffer.py:39-44,72-130) on the same
    state; draws follow the probabilities."""
    from muzero_amd.replay_buffer import ReplayBuffer
    store = DeviceEpisodeStore(games=6, max_steps=40, obs_shape=OBS, num_actions=A, capacity=64, device="cpu")
    _play(store, 120, seed=3)
    n = store.size
    assert n >= 12
    eps = store.to_episodes()
    host = ReplayBuffer(buffer_size=64, batch_size=16, alpha=0.6, beta=0.4)
    host.add_episodes(eps)
    assert torch.equal(store.priority[:n], torch.ones(n, dtype=torch.float64))
    # the same updates on both sides, slot 2 twice
    slots = torch.tensor([2, 5, 2, 0, 7])
    errors = torch.tensor([0.5, -3.0, 0.25, 1e-9, 2.0])
    store.update_priorities(slots, errors)
    host.update_priorities(slots.tolist(), errors.tolist())
    want_pri = np.array([p for _, p in host.episode_buffer])
    np.testing.assert_allclose(store.priority[:n].numpy(), want_pri, rtol=0, atol=0)
    assert float(store.max_priority) == host.max_priority == 3.0 + 1e-6
    scaled = want_pri ** 0.6
    np.testing.assert_allclose(store.sampling_probabilities(0.6).numpy(), scaled / scaled.sum(), rtol=1e-14)
    # a draw: weights as the host computes them for the same indices; short episodes dropped
    gen = torch.Generator().manual_seed(0)
    batch, got_slots, k = store.sample_prioritized(16, lookback=2, roll_ahead=3, history_length=2, discount=0.99,
                                                   z_targets=_z, alpha=0.6, beta=0.4, generator=gen)
    probs = scaled / scaled.sum()
    gen = torch.Generator().manual_seed(0)
    draw = torch.multinomial(torch.from_numpy(probs), 16, replacement=True, generator=gen)
    w = (n * probs[draw.numpy()]) ** -0.4
    w = w / w.max()
    keep = store.done_len[draw].numpy() >= 2 + 3 + 1
    assert 0 < keep.sum() < 16  # the draw holds episodes of both kinds
    # dropped draws stay in the batch with weight 0 and the scratch slot
    assert torch.equal(got_slots, torch.where(torch.from_numpy(keep), draw, torch.tensor(store.cap)))
    np.testing.assert_allclose(batch.is_weights.numpy(), np.where(keep, w, 0.0).astype(np.float32), rtol=1e-6)
    cut_from = torch.where(torch.from_numpy(keep), draw, torch.argmax(store.done_len[:n]))
    lens = store.done_len[cut_from]
    assert bool(((k >= 2) & (k <= lens - 3 - 1)).all()) and len(batch) == 16
    # and they do not touch the priorities
    before, top = store.priority.clone(), float(store.max_priority)
    store.update_priorities(got_slots[~torch.from_numpy(keep)], torch.full((int((~keep).sum()),), 99.0))
    assert torch.equal(store.priority, before) and float(store.max_priority) == top
    # new episodes enter with the largest priority so far
    _play(store, 30, seed=4)
    newest = (store.n_finished - 1) % store.cap
    assert float(store.priority[newest]) == 3.0 + 1e-6
    # draw frequencies follow the probabilities (4 sigma)
    probs = store.sampling_probabilities(1.0)
    gen = torch.Generator().manual_seed(1)
    counts = torch.zeros(store.size)
    for _ in range(40):
        d = torch.multinomial(probs, 64, replacement=True, generator=gen)
        counts += torch.bincount(d, minlength=store.size)
    N = 40 * 64
    sigma = torch.sqrt(N * probs * (1 - probs))
    assert bool(((counts - N * probs).abs() <= 4 * sigma + 1).all())
