"""GPU parity of the data-side kernels: history ring/tensor and z-targets bit-exact against
the reference-generated vectors and the CPU oracle; the 1x1-conv representation network and the
action pick against torch."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import lib as oracle_lib
from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu
DATA = np.load(os.path.join(GOLDEN, "data_path.npz"))


def test_history_kernel_matches_reference_vectors():
    """Two games run the recorded sequences with a one-step lag between them (different ring
    phases in one launch); every step's history tensor must equal the reference's bit for bit."""
    from muzero_amd.selfplay import DeviceHistory
    hi0 = 0
    for K in (1, 3, 4):
        frames, actions = DATA[f"histseq_K{K}_frames"][:, 0], DATA[f"histseq_K{K}_actions"]
        T, shape = frames.shape[0], frames.shape[1:]
        hist = DeviceHistory(2, K, tuple(shape))
        for t in range(T + 1):
            t0, t1 = min(t, T - 1), max(t - 1, 0)  # game 1 lags one step
            fr = torch.from_numpy(np.stack([frames[t0], frames[t1]])).cuda().contiguous()
            ac = torch.from_numpy(np.array([actions[t0], actions[t1]], np.int32)).cuda()
            st = torch.tensor([t0 if t < T else T, t1 if t > 0 else 0], dtype=torch.int32, device="cuda")
            if t == T:
                st[0] = T - 1  # game 0 idles: re-pushing the same step is idempotent
            out = hist.push(fr, st, ac).cpu().numpy()
            if t < T:
                assert np.array_equal(out[0:1], DATA[f"hist{hi0 + t}_out"]), (K, t)
            if t > 0:
                assert np.array_equal(out[1:2], DATA[f"hist{hi0 + t - 1}_out"]), (K, t)
        hi0 += T


def test_z_target_kernel_bit_exact():
    from muzero_amd.training import compute_z_targets_batch
    for i in range(int(DATA["n_z"])):
        w, gamma = int(DATA[f"z{i}_w"]), float(DATA[f"z{i}_gamma"])
        z = compute_z_targets_batch(DATA[f"z{i}_R"][None], DATA[f"z{i}_V"][None], gamma)
        assert np.array_equal(z[0], DATA[f"z{i}_z"])
    # a large random batch against the CPU oracle
    rng = np.random.default_rng(0)
    B, w, gamma = 4096, 5, 0.997
    R, V = rng.standard_normal((B, w + 1)), rng.standard_normal((B, w + 1))
    z = compute_z_targets_batch(R, V, gamma)
    want = np.zeros_like(z)
    dp = C.POINTER(C.c_double)
    for b in range(B):
        oracle_lib().mzo_z_targets(np.ascontiguousarray(R[b]).ctypes.data_as(dp), np.ascontiguousarray(V[b]).ctypes.data_as(dp),
                                   C.c_int32(w), C.c_double(gamma), want[b].ctypes.data_as(dp))
    assert np.array_equal(z, want)


def test_pointwise_representation_matches_torch_and_reference_vectors():
    from muzero_amd.selfplay import RepresentationRunner
    from tests.test_gpu_nets import _c4_nets
    h, _, _ = _c4_nets(0)
    run = RepresentationRunner(h)
    assert run.pointwise
    z = np.load(os.path.join(GOLDEN, "c4_mlp.npz"))
    out = run(torch.from_numpy(z["h_in"]).cuda()).cpu().numpy()
    np.testing.assert_allclose(out, z["h_out"], rtol=1e-5, atol=1e-6)
    x = torch.randn(300, 3, 6, 7)
    with torch.no_grad():
        np.testing.assert_allclose(run(x.cuda()).cpu().numpy(), h(x).numpy(), rtol=1e-5, atol=1e-6)


def test_pick_action_first_argmax_and_exploration():
    from muzero_amd.selfplay import pick_action
    G, A = 1000, 7
    rng = np.random.default_rng(2)
    pol = rng.random((G, A)).astype(np.float32)
    pol[::3, 2] = pol[::3, 5] = 2.0  # exact ties: the first maximum must win
    explore = (rng.random(G) < 0.3).astype(np.int8)
    rnd = rng.integers(0, A, size=G).astype(np.int32)
    out = torch.zeros(G, dtype=torch.int32, device="cuda")
    pick_action(torch.from_numpy(pol).cuda(), torch.from_numpy(explore).cuda(), torch.from_numpy(rnd).cuda(), out)
    want = np.where(explore == 1, rnd, torch.from_numpy(pol).argmax(dim=1).numpy())
    assert np.array_equal(out.cpu().numpy(), want)
    assert (want[::3][explore[::3] == 0] == 2).all()


def test_move_draws_distributions_and_streams():
    """mz_move_draws: the four random inputs of a move from one launch.  Same distributions as the reference's
    draws (src/search/mcts.py:64-66 with expansion.py:35: uniform child; mcts.py:85-94: Dirichlet(alpha);
    src/training_data_generator.py:148-163: coin against epsilon, action ~ random_action_probability), streams
    a function of (seed, move number, game): reproducible, advancing by themselves, independent of the batch size."""
    import numpy as np
    from muzero_amd import _hip
    from muzero_amd._hip import ptr
    lib = _hip.lib()
    G, A, S1, alpha = 20000, 7, 51, 0.3
    dev = "cuda"
    p_rand = torch.tensor([0.05, 0.1, 0.15, 0.2, 0.25, 0.15, 0.1], device=dev)

    def draw(state, eps, n=G, p=p_rand, seed=1234):
        e = torch.tensor([eps], dtype=torch.float32, device=dev)
        out = (torch.empty((n, S1), dtype=torch.int32, device=dev), torch.empty((n, A), device=dev),
               torch.empty(n, dtype=torch.int8, device=dev), torch.empty(n, dtype=torch.int32, device=dev))
        _hip.check(lib.mz_move_draws(seed, ptr(state), ptr(e), ptr(p), alpha, n, A, S1, *[ptr(t) for t in out], None), "mz_move_draws")
        torch.cuda.synchronize()
        return out

    state = torch.zeros(2, dtype=torch.int64, device=dev)
    rand, eta, explore, rnd = draw(state, 0.25)
    assert state.tolist() == [1, 0]
    # uniform children: every action within 4 sigma of G * S1 / A
    counts = torch.bincount(rand.reshape(-1).long(), minlength=A).double().cpu().numpy()
    n = G * S1
    assert counts.sum() == n and np.abs(counts - n / A).max() < 4 * np.sqrt(n / A)
    assert int(rand.min()) >= 0 and int(rand.max()) < A
    # Dirichlet(alpha): rows on the simplex, component mean 1/A, variance (1/A)(1 - 1/A) / (A alpha + 1)
    e64 = eta.double().cpu().numpy()
    np.testing.assert_allclose(e64.sum(1), 1.0, atol=1e-5)
    assert (e64 >= 0).all()
    var = (1 / A) * (1 - 1 / A) / (A * alpha + 1)
    assert np.abs(e64.mean(0) - 1 / A).max() < 5 * np.sqrt(var / G)
    np.testing.assert_allclose(e64.var(0), var, rtol=0.06)
    # a sparse prior: the share of components below 1e-3 equals the Beta(alpha, (A - 1) alpha) mass there
    from scipy import stats
    want = stats.beta(alpha, (A - 1) * alpha).cdf(1e-3)
    assert abs((e64 < 1e-3).mean() - want) < 0.01
    # coin and random action
    assert abs(float(explore.float().mean()) - 0.25) < 4 * np.sqrt(0.25 * 0.75 / G)
    pc = torch.bincount(rnd.long(), minlength=A).double().cpu().numpy() / G
    assert np.abs(pc - p_rand.cpu().numpy()).max() < 4 * np.sqrt(0.25 / G)
    # the next call draws the next move; a fresh state with the same seed repeats the first; another seed differs
    rand2, eta2, _, _ = draw(state, 0.25)
    assert state.tolist() == [2, 0] and not torch.equal(rand2, rand) and not torch.equal(eta2, eta)
    again = draw(torch.zeros(2, dtype=torch.int64, device=dev), 0.25)
    assert all(torch.equal(a, b) for a, b in zip(again, (rand, eta, explore, rnd)))
    other = draw(torch.zeros(2, dtype=torch.int64, device=dev), 0.25, seed=99)
    assert not torch.equal(other[0], rand)
    # a game's stream does not depend on how many games are drawn with it; uniform action when no table is given
    few = draw(torch.zeros(2, dtype=torch.int64, device=dev), 0.25, n=300)
    assert torch.equal(few[0], rand[:300]) and torch.equal(few[1], eta[:300])
    uni = draw(torch.zeros(2, dtype=torch.int64, device=dev), 1.0, p=None)
    assert bool(uni[2].all())
    pu = torch.bincount(uni[3].long(), minlength=A).double().cpu().numpy() / G
    assert np.abs(pu - 1 / A).max() < 4 * np.sqrt((1 / A) / G)


def test_episode_end_counts_steps_and_ends_games():
    """mz_episode_end against the torch expressions it replaces (the episode loop of
    src/training_data_generator.py:107-142: a game ends when its environment says so or at the step limit)."""
    from muzero_amd import _hip
    from muzero_amd._hip import ptr
    G, limit = 1000, 42
    gen = torch.Generator(device="cuda").manual_seed(3)
    step = torch.randint(0, limit, (G,), device="cuda", generator=gen, dtype=torch.int32)
    done = (torch.rand(G, device="cuda", generator=gen) < 0.2).to(torch.int8)
    want_step = step + 1
    want_end = (done.bool() | (want_step >= limit)).to(torch.int8)
    want_step = want_step.masked_fill(want_end.bool(), 0)
    ended = torch.full((G,), 7, dtype=torch.int8, device="cuda")
    _hip.check(_hip.lib().mz_episode_end(ptr(done), ptr(step), limit, G, ptr(ended), None), "mz_episode_end")
    torch.cuda.synchronize()
    assert torch.equal(ended, want_end) and torch.equal(step, want_step)
    assert int(want_end.sum()) > 0 and int((want_step == 0).sum()) >= int(want_end.sum())
