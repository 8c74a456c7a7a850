"""Batched / data-parallel learner step (CPU): the batched unroll gives the loss and the
gradient of the per-window B=1 loop that mirrors src/training.py:131-172,244-305, and two gloo
ranks that split the windows end at the weights one rank computes from all of them."""
import copy
import os
import random
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

A, W, LOOKBACK = 7, 5, 4


def _nets(seed=0):
    from muzero_amd.config import load_config
    from muzero_amd.neural_networks import DynamicsNetwork, PredictionNetwork, RepresentationNetwork
    cfg = load_config("config_c4_mlp.yaml")
    torch.manual_seed(seed)
    h = RepresentationNetwork((2, 6, 7), cfg.networks.latent_shape, cfg.networks.representation)
    g = DynamicsNetwork(cfg.networks.latent_shape, A, cfg.networks.dynamics)
    f = PredictionNetwork(cfg.networks.latent_shape, A, cfg.networks.prediction)
    return cfg, h, g, f


def _episodes(n, length, seed):
    from muzero_amd.training_data_generator import Chunk, Episode
    rng = np.random.default_rng(seed)
    eps = []
    for _ in range(n):
        chunks = []
        for _ in range(length):
            p = rng.random(A).astype(np.float32)
            chunks.append(Chunk(state=torch.from_numpy(rng.integers(0, 2, size=(1, 2, 6, 7)).astype(np.float32)),
                                policy=torch.from_numpy(p / p.sum()), reward=float(rng.normal()), value=float(rng.normal()),
                                best_action=int(rng.integers(0, A))))
        eps.append(Episode(chunks=chunks))
    return eps


def _windows(seed, n=6):
    rnd = random.Random(seed)
    eps = _episodes(n, 16, seed)
    wins = [(ep, rnd.randrange(LOOKBACK, len(ep.chunks) - W)) for ep in eps]
    R = np.array([[ep.chunks[i].reward for i in range(k, k + W + 1)] for ep, k in wins], np.float64)
    V = np.array([[ep.chunks[i].value for i in range(k, k + W + 1)] for ep, k in wins], np.float64)
    isw = np.random.default_rng(seed).random(n) + 0.5
    return wins, R, V, isw


def _reference_style_loss(mgr, wins, Z, R, isw):
    """The per-window loop of NeuralNetworkManager.train, without the optimizer step."""
    total = torch.zeros(1)
    for row, (ep, k) in enumerate(wins):
        start = max(0, k - LOOKBACK)
        past_s = [ep.chunks[i].state for i in range(start, k)]
        past_a = [ep.chunks[i].best_action for i in range(start, k)]
        acts = [ep.chunks[i].best_action for i in range(k, k + W + 1)]
        pols = [ep.chunks[i].policy for i in range(k, k + W + 1)]
        p, v, r = mgr.bptt(past_s, past_a, acts, (pols, [float(z) for z in Z[row]], [float(x) for x in R[row]]))
        total = total + float(isw[row]) * (p + v + r)
    return total / float(np.sum(isw))


def _z_targets(R, V, g=0.998):
    """z_k = sum_i g^(i-1) R[k+i] + g^(w-k) V[w] (src/training.py:223-242, first term g^-1)."""
    w = R.shape[1] - 1
    Z = np.zeros_like(R)
    for k in range(w + 1):
        Z[:, k] = sum(g ** (i - 1) * R[:, k + i] for i in range(w - k + 1)) + g ** (w - k) * V[:, w]
    return Z


def test_batched_unroll_equals_per_window_loop():
    from muzero_amd.learner import batched_unroll_loss, windows_to_batch
    from muzero_amd.training import NeuralNetworkManager
    cfg, h, g, f = _nets()
    mgr = NeuralNetworkManager(cfg.training, h, g, f)
    wins, R, V, isw = _windows(3)
    Z = _z_targets(R, V)
    want = _reference_style_loss(mgr, wins, Z, R, isw)
    for n in (h, g, f):
        n.zero_grad()
    want.backward()
    want_grads = [p.grad.clone() for n in (h, g, f) for p in n.parameters()]
    for n in (h, g, f):
        n.zero_grad()
    batch = windows_to_batch(wins, Z, R, isw, h.history_length, LOOKBACK, W)
    wsum, per_window, _ = batched_unroll_loss(h, g, f, batch, W)
    got = wsum / batch.is_weights.sum()
    got.backward()
    assert abs(got.item() - want.item()) <= 1e-5 * abs(want.item())
    assert per_window.shape == (len(wins),)
    for p, wg in zip([p for n in (h, g, f) for p in n.parameters()], want_grads):
        torch.testing.assert_close(p.grad, wg, rtol=2e-4, atol=2e-6)


def _dp_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from muzero_amd.learner import DataParallelLearner, windows_to_batch
        cfg, h, g, f = _nets()
        params = [p for n in (h, g, f) for p in n.parameters()]
        opt = torch.optim.Adam(params, lr=1e-3)
        dp = DataParallelLearner(h, g, f, opt, W, dist=dist)
        losses = []
        for step in range(2):
            wins, R, V, isw = _windows(10 + step, n=5)  # 5 windows over 2 ranks: uneven shares
            batch = windows_to_batch(wins, _z_targets(R, V), R, isw, h.history_length, LOOKBACK, W)
            loss, errs, mine = dp.step(batch)
            assert list(mine) == list(range(rank, 5, world)) and len(errs) == len(mine)
            losses.append(loss)
        if rank == 1:  # knock rank 1 out of step, then let the learner's broadcast repair it
            with torch.no_grad():
                params[0].add_(1.0)
        dp.broadcast_weights(src=0)
        torch.save({"losses": losses, "params": [p.detach().clone() for p in params]}, os.path.join(tmp, f"dp{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_data_parallel_step_equals_single_rank(tmp_path):
    from muzero_amd.learner import DataParallelLearner, windows_to_batch
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_dp_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    cfg, h, g, f = _nets()
    params = [p for n in (h, g, f) for p in n.parameters()]
    opt = torch.optim.Adam(params, lr=1e-3)
    single = DataParallelLearner(h, g, f, opt, W)
    losses = []
    for step in range(2):
        wins, R, V, isw = _windows(10 + step, n=5)
        batch = windows_to_batch(wins, _z_targets(R, V), R, isw, h.history_length, LOOKBACK, W)
        losses.append(single.step(batch)[0])
    r0 = torch.load(tmp_path / "dp0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "dp1.pt", weights_only=True)
    np.testing.assert_allclose(r0["losses"], losses, rtol=1e-5)
    np.testing.assert_allclose(r1["losses"], losses, rtol=1e-5)
    for a, b, c in zip(r0["params"], r1["params"], params):
        assert torch.equal(a, b)  # ranks in step after the broadcast
        torch.testing.assert_close(a, c.detach(), rtol=1e-4, atol=1e-6)


def test_batched_unroll_with_residual_towers_equals_per_window_loop():
    """Networks with BatchNorm (conv h with a ResBlock, residual-tower g / f): the batched unroll
    normalises every window with its own statistics, so it still equals the reference's loop of
    batch-size-1 unrolls in loss and gradient."""
    from muzero_amd.config import (ConvLayerConfig, DenseLayerConfig, DynamicsNetworkConfig, PredictionNetworkConfig,
                                   RepresentationNetworkConfig, ResBlockConfig, load_config)
    from muzero_amd.learner import batched_unroll_loss, windows_to_batch
    from muzero_amd.neural_networks import DynamicsNetwork, PredictionNetwork, RepresentationNetwork
    from muzero_amd.training import NeuralNetworkManager
    cfg = load_config("config_c4_mlp.yaml")
    rb = lambda c: ResBlockConfig(out_channels=c, kernel_size=3, stride=1, padding=1, activation="relu",  # noqa: E731
                                  pool_kernel_size=1, pool_stride=1)
    d = lambda n: DenseLayerConfig(out_features=n, activation="relu")  # noqa: E731
    torch.manual_seed(5)
    shape = (5, 6, 7)
    h = RepresentationNetwork((2, 6, 7), shape, RepresentationNetworkConfig(
        history_length=cfg.networks.representation.history_length,
        downsample=[ConvLayerConfig(out_channels=6, kernel_size=3, stride=1, padding=1, activation="relu"), rb(6)],
        res_net=[rb(4)]))
    g = DynamicsNetwork(shape, A, DynamicsNetworkConfig(res_net=[rb(3)], reward_net=[d(16)]))
    f = PredictionNetwork(shape, A, PredictionNetworkConfig(res_net=[rb(3)], value_net=[d(16)], policy_net=[d(16)]))
    mgr = NeuralNetworkManager(cfg.training, h, g, f)
    wins, R, V, isw = _windows(4)
    Z = _z_targets(R, V)
    want = _reference_style_loss(mgr, wins, Z, R, isw)
    for n in (h, g, f):
        n.zero_grad()
    want.backward()
    params = [p for n in (h, g, f) for p in n.parameters()]
    want_grads = [p.grad.clone() for p in params]
    for n in (h, g, f):
        n.zero_grad()
    batch = windows_to_batch(wins, Z, R, isw, h.history_length, LOOKBACK, W)
    wsum, per_window, _ = batched_unroll_loss(h, g, f, batch, W)
    got = wsum / batch.is_weights.sum()
    got.backward()
    assert abs(got.item() - want.item()) <= 2e-5 * abs(want.item())
    for p, wg in zip(params, want_grads):
        torch.testing.assert_close(p.grad, wg, rtol=2e-3, atol=2e-5)
    assert all("forward" not in m.__dict__ for n in (h, g, f) for m in n.modules())


def test_zero_weight_batch_leaves_weights_unchanged_and_finite():
    """A draw made only of episodes too short for a window arrives as zero-weight windows
    (DeviceEpisodeStore.sample_prioritized keeps the batch shape): the weight total is 0 and the
    reference skips the optimizer step (`if ep_indices_used:`, src/training.py:174)."""
    from muzero_amd.learner import DataParallelLearner, windows_to_batch
    cfg, h, g, f = _nets()
    params = [p for n in (h, g, f) for p in n.parameters()]
    opt = torch.optim.SGD(params, lr=0.1, momentum=0.9, weight_decay=1e-2)
    dp = DataParallelLearner(h, g, f, opt, W)
    wins, R, V, isw = _windows(21, n=4)
    good = windows_to_batch(wins, _z_targets(R, V), R, isw, h.history_length, LOOKBACK, W)
    dp.step_local(good)  # momentum buffers exist and are non-zero now
    before = [p.detach().clone() for p in params]
    dropped = windows_to_batch(wins, _z_targets(R, V), R, np.zeros(4), h.history_length, LOOKBACK, W)
    loss, errs = dp.step_local(dropped)
    assert loss == 0.0 and np.isfinite(errs.numpy()).all()
    for p, b in zip(params, before):
        assert torch.isfinite(p).all() and torch.equal(p.detach(), b)
    loss2, _ = dp.step_local(good)  # and training goes on
    assert np.isfinite(loss2) and all(torch.isfinite(p).all() for p in params)
