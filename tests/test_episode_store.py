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
    store = DeviceEpisodeStore(4, 6, OBS, A, capacity=3, device="cpu")
    _play(store, 40, seed=9)  # max_steps 6: every episode is cut at 6 moves at the latest
    assert store.size == 3 and store.n_finished > 3
    assert (store.done_len <= 6).all()
    assert store.eligible(4, 5).numel() == 0  # needs 10 chunks
    with pytest.raises(ValueError):
        store.sample_windows(2, 4, 5, 1, 0.99, _z)
