"""The replay kernels (mz_per_sample, mz_per_update, mz_replay_cut_windows) through a CUDA episode
store: on numpy's own uniform stream the draw equals ReplayBuffer.sample_batch's np.random.choice
(src/replay_buffer.py:72-119) index for index, weights and probabilities to 1e-12 (device pow against
the C library's), priority updates exactly; windows equal the list-based path bit for bit
(src/training.py:131-160, 223-262) for history lengths 1..4."""
import numpy as np
import pytest
import torch

from muzero_amd.episode_store import DeviceEpisodeStore
from muzero_amd.learner import windows_to_batch
from tests.test_episode_store import A, OBS, _play, _z

pytestmark = pytest.mark.gpu


def _store(games, max_steps, capacity, moves, seed):
    """The same random games recorded into a CPU store (the host-side definition) and a CUDA store."""
    cpu = DeviceEpisodeStore(games, max_steps, OBS, A, capacity=capacity, device="cpu")
    gpu = DeviceEpisodeStore(games, max_steps, OBS, A, capacity=capacity, device="cuda")
    _play(cpu, moves, seed)

    class Both:
        G = games

        def record(self, *a):
            gpu.record(*(t.cuda() for t in a))
    _play(Both(), moves, seed)
    return cpu, gpu


@pytest.mark.parametrize("K", [1, 2, 3, 4])
def test_windows_from_the_kernel_equal_the_list_based_path(K):
    lookback, w, gamma = 4, 5, 0.998
    cpu, gpu = _store(12, 30, 64, 80, seed=K)
    assert gpu.size == cpu.size > 10
    episodes = gpu.to_episodes()
    gen = torch.Generator(device="cuda").manual_seed(5)
    batch, slots, k = gpu.sample_windows(24, lookback, w, K, gamma, None, generator=gen)
    slots, k = slots.cpu(), k.cpu()
    assert ((k >= lookback) & (k + w < cpu.done_len[slots])).all()
    wins = [(episodes[int(s)], int(kk)) for s, kk in zip(slots, k)]
    R = np.array([[ep.chunks[i].reward for i in range(kk, kk + w + 1)] for ep, kk in wins], np.float64)
    V = np.array([[ep.chunks[i].value for i in range(kk, kk + w + 1)] for ep, kk in wins], np.float64)
    from muzero_amd.training import compute_z_targets_batch
    want = windows_to_batch(wins, compute_z_targets_batch(R, V, gamma), R, np.ones(len(wins)), K, lookback, w)
    assert torch.equal(batch.history.cpu(), want.history)
    assert torch.equal(batch.actions.cpu(), want.actions) and torch.equal(batch.policies.cpu(), want.policies)
    assert torch.equal(batch.z.cpu(), want.z) and torch.equal(batch.rewards.cpu(), want.rewards)
    # the host-side formula of the targets agrees as well (libm pow against the table: 1e-12)
    np.testing.assert_allclose(batch.z.cpu().numpy(), _z(torch.from_numpy(R), torch.from_numpy(V), gamma).numpy(), rtol=1e-6)


def test_prioritised_draw_equals_numpy_choice_on_the_same_stream():
    from muzero_amd.replay_buffer import ReplayBuffer
    cpu, gpu = _store(6, 40, 64, 160, seed=3)
    n = gpu.size
    assert n >= 12
    host = ReplayBuffer(buffer_size=64, batch_size=16, alpha=0.6, beta=0.4)
    host.add_episodes(gpu.to_episodes())
    # identical priority updates on both sides (slot 2 twice: the last one wins), several rounds
    rng = np.random.default_rng(0)
    for rnd in range(3):
        slots = torch.tensor([2, 5, 2, 0, 7, n - 1]) if rnd == 0 else torch.from_numpy(rng.integers(0, n, size=9))
        errors = torch.tensor([0.5, -3.0, 0.25, 1e-9, 2.0, 0.75]) if rnd == 0 else torch.from_numpy(rng.normal(size=9).astype(np.float32))
        gpu.update_priorities(slots.cuda(), errors.cuda())
        host.update_priorities(slots.tolist(), [float(e) for e in errors])
        want_pri = np.array([p for _, p in host.episode_buffer])
        assert np.array_equal(gpu.priority[:n].cpu().numpy(), want_pri)
        assert float(gpu.max_priority) == host.max_priority
        # the same numpy stream on both sides: ReplayBuffer draws with np.random.choice ...
        np.random.seed(100 + rnd)
        _, want_idx, want_w = host.sample_batch()
        # ... which consumes random_sample(16); the store is handed those numbers
        np.random.seed(100 + rnd)
        u = np.random.random_sample(16)
        u_start = np.random.default_rng(rnd).random(16)
        batch, got_slots, k = gpu.sample_prioritized(16, lookback=2, roll_ahead=3, history_length=2, discount=0.99,
                                                     z_targets=None, alpha=0.6, beta=0.4, uniforms=(u, u_start))
        assert gpu.last_draw.cpu().tolist() == [int(i) for i in want_idx]
        np.testing.assert_allclose(gpu.last_weights.cpu().numpy(), want_w, rtol=1e-12)
        scaled = want_pri ** 0.6
        np.testing.assert_allclose(gpu.last_probs.cpu().numpy(), scaled / scaled.sum(), rtol=1e-12)
        # window choice: short episodes dropped (weight 0, scratch slot), starts inside the admissible range
        lens = cpu.done_len[:n].numpy()
        keep = lens[np.array(want_idx)] >= 2 + 3 + 1
        assert got_slots.cpu().tolist() == [int(i) if kp else gpu.cap for i, kp in zip(want_idx, keep)]
        np.testing.assert_allclose(batch.is_weights.cpu().numpy(), np.where(keep, want_w, 0.0).astype(np.float32), rtol=1e-6)
        cut_from = np.where(keep, want_idx, int(np.argmax(lens)))
        span = lens[cut_from] - 3 - 2
        assert k.cpu().tolist() == [2 + min(int(us * sp), sp - 1) for us, sp in zip(u_start, span)]
        assert len(batch) == 16 and batch.history.shape[1] == OBS[0] * 2 + 2
    assert 0 < keep.sum()
    # uniform fallback when every priority vanishes (src/replay_buffer.py:92-94)
    gpu.priority.zero_()
    u = np.linspace(0.01, 0.99, 8)
    gpu.sample_prioritized(8, 2, 3, 2, 0.99, None, alpha=0.6, beta=0.4, uniforms=(u, u))
    np.testing.assert_allclose(gpu.last_probs.cpu().numpy(), np.full(n, 1.0 / n), rtol=0)
    assert gpu.last_draw.cpu().tolist() == [int(np.searchsorted(np.cumsum(np.full(n, 1.0 / n)) / np.cumsum(np.full(n, 1.0 / n))[-1], x, side="right")) for x in u]
