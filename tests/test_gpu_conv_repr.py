"""The convolutional representation network h on the HIP kernels (muzero_amd.conv_repr) against the
torch module evaluated as the reference evaluates it: ONE game per call, train mode (per-call
BatchNorm statistics), src/training_data_generator.py:117 with
src/neural_networks/neural_network.py:83-102 and network_builder.py:32-151.

Tolerance: 1e-5 relative with an absolute floor of 1e-5 of the output's scale.  The float32 torch
reference is itself a rounded computation; where the shapes allow it the float64 evaluation of the
same module is the yardstick and the float32 torch result is shown to be no closer to it than ours.
"""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _net(obs, latent, K, downsample, res_net, seed=0):
    from muzero_amd.config import RepresentationNetworkConfig
    from muzero_amd.neural_networks import RepresentationNetwork
    torch.manual_seed(seed)
    h = RepresentationNetwork(obs, latent, RepresentationNetworkConfig(history_length=K, downsample=downsample, res_net=res_net))
    with torch.no_grad():
        for m in h.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
    return h


def _cv(c, k, s, p, act="relu"):
    from muzero_amd.config import ConvLayerConfig
    return ConvLayerConfig(out_channels=c, kernel_size=k, stride=s, padding=p, activation=act)


def _rb(c, k=3, s=1, p=1, act="relu", pk=1, ps=1):
    from muzero_amd.config import ResBlockConfig
    return ResBlockConfig(out_channels=c, kernel_size=k, stride=s, padding=p, activation=act, pool_kernel_size=pk, pool_stride=ps)


def _pool(k, s, kind):
    from muzero_amd.config import PoolLayerConfig
    return PoolLayerConfig(kernel_size=k, stride=s, pool_type=kind)


def _per_game(h, x, dtype=torch.float32, device="cpu"):
    m = copy.deepcopy(h).to(device=device, dtype=dtype)
    m.train()
    with torch.no_grad():
        out = torch.cat([m(x[i:i + 1].to(device=device, dtype=dtype)) for i in range(x.shape[0])])
    return out, m


def _close(got, want, tol=1e-5):
    want = want.double().cpu().numpy()
    np.testing.assert_allclose(got.double().cpu().numpy(), want, rtol=tol, atol=tol * float(np.abs(want).max()))


@pytest.mark.parametrize("fast", [True, False], ids=["split-fp16", "fp32-only"])
@pytest.mark.parametrize("case", ["direct-small", "matrix-64", "mixed-strides", "avgpool-tanh", "pointwise-pooled", "wide-128"])
def test_conv_representation_matches_torch_per_game(case, fast):
    from muzero_amd.conv_repr import ConvRepresentation
    if case == "direct-small":      # few channels everywhere: the direct kernel; identity-skip and 1x1-skip blocks
        h = _net((2, 8, 8), (5, 4, 4), 2, [_cv(6, 2, 2, 0), _rb(6), _rb(4)], [_rb(4, k=1, p=0)])
        G, obs = 5, (2, 8, 8)
    elif case == "matrix-64":       # 64-channel layers: gathered matrix products (3x3, 2x2 stride 2, 1x1 skip, 1x1 final)
        h = _net((3, 12, 12), (16, 3, 3), 16, [_cv(64, 2, 2, 0), _rb(64), _cv(128, 2, 2, 0), _rb(128), _rb(3)], [])
        G, obs = 4, (3, 12, 12)
    elif case == "mixed-strides":   # a strided block with pooling, 5x5 kernels, a block that changes the channel count
        h = _net((1, 20, 20), (8, 2, 2), 3, [_cv(64, 3, 1, 1, act="tanh"), _rb(64, k=3, s=2, p=1, pk=2, ps=2), _rb(32, k=5, p=2)], [_rb(8)])
        G, obs = 3, (1, 20, 20)
    elif case == "avgpool-tanh":    # average pooling layer, sigmoid / tanh activations, non-square-divisible adaptive pool
        h = _net((2, 14, 14), (6, 3, 3), 1, [_cv(8, 3, 1, 1, act="sigmoid"), _pool(2, 2, "avg"), _rb(6, act="tanh")], [])
        G, obs = 4, (2, 14, 14)
    elif case == "wide-128":        # 128 / 256-channel layers: the split-fp16 product everywhere (3x3, 2x2 stride 2, 1x1 skip, tanh)
        h = _net((2, 16, 16), (256, 2, 2), 31, [_cv(128, 2, 2, 0), _rb(128), _rb(256, act="tanh"), _cv(256, 3, 1, 1, act="sigmoid"),
                                                 _rb(256, k=3, s=2, p=1)], [])
        G, obs = 3, (2, 16, 16)
    else:                           # a point-wise network whose latent is smaller than the observation: 1x1 conv + adaptive pool
        h = _net((2, 6, 7), (4, 3, 3), 1, [], [])
        G, obs = 6, (2, 6, 7)
    K = h.history_length
    torch.manual_seed(1)
    x = torch.rand(G, obs[0] * K + K, obs[1], obs[2])
    want64, _ = _per_game(h, x, torch.float64)
    want32, _ = _per_game(h, x, torch.float32)
    run = ConvRepresentation(h, fast_convolutions=fast)
    if case == "wide-128":
        assert any(c.fast for c in run.convs) == fast
    got = run(x.cuda())
    torch.cuda.synchronize()
    assert got.shape == want64.shape
    _close(got, want64)
    # our distance from the float64 result is of the order of float32 torch's own
    scale = float(want64.abs().max())
    ours = float((got.double().cpu() - want64).abs().max()) / scale
    theirs = float((want32.double() - want64).abs().max()) / scale
    assert ours <= max(4 * theirs, 2e-6), (ours, theirs)
    # a batch must NOT share statistics: the batched torch call differs from the per-game result
    if any(isinstance(m, torch.nn.BatchNorm2d) for m in h.modules()):
        with torch.no_grad():
            batched = copy.deepcopy(h).train()(x)
        assert float((batched - want32).abs().max()) > 1e-3 * scale


def test_running_statistics_follow_the_reference_calls():
    """track_running_stats: after G games the BatchNorm buffers equal those the reference's G
    batch-size-1 train-mode calls leave (network_builder.py:94,104,118; momentum 0.1, unbiased
    variance), twice in a row, and sync_running_stats writes them into the caller's module."""
    from muzero_amd.conv_repr import ConvRepresentation
    h = _net((3, 12, 12), (16, 3, 3), 4, [_cv(64, 2, 2, 0), _rb(64), _rb(3)], [_rb(3)])
    G = 7
    K = h.history_length
    torch.manual_seed(2)
    xs = [torch.rand(G, 3 * K + K, 12, 12) for _ in range(2)]
    ref = copy.deepcopy(h).train()
    run = ConvRepresentation(h, track_running_stats=True)
    for x in xs:
        with torch.no_grad():
            for i in range(G):
                ref(x[i:i + 1])
        run(x.cuda())
    run.sync_running_stats()
    for (name, a), (_, b) in zip(h.named_buffers(), ref.named_buffers()):
        if name.endswith("num_batches_tracked"):
            assert int(a) == int(b) == 2 * G, name
        else:
            np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-5, atol=1e-6, err_msg=name)
    # the parameters themselves were not touched, so the evaluator did not need to re-read them
    assert all(torch.equal(p, q) for p, q in zip(h.parameters(), ref.parameters()))


def test_running_statistics_across_selfplay_training_selfplay():
    """Self-play, a training step, self-play again, through the runner self-play uses.  In the reference's order
    (self-play's buffers handed over before training, as SelfPlayEngine.generate does at its end) the module's buffers
    equal those of a torch twin given the same calls in the same order.  Without the hand-over, the buffers a training
    step left are NOT overwritten by the stale self-play copy when the runner re-reads the changed weights: the
    training step's values stay, a warning says that self-play's updates were dropped."""
    import warnings
    from muzero_amd.selfplay import RepresentationRunner
    h = _net((3, 12, 12), (16, 3, 3), 2, [_cv(64, 2, 2, 0), _rb(64), _rb(3)], [_rb(3)])
    G, K = 5, 2
    torch.manual_seed(6)
    x1, x2, xb = (torch.rand(n, 3 * K + K, 12, 12) for n in (G, G, 4))

    def train_step(m):  # a train-mode batch call (advances the buffers) and an in-place parameter update
        m.train()
        with torch.no_grad():
            m(xb)
            for p in m.parameters():
                p.mul_(0.999)

    def buffers_equal(a, b):
        for (name, u), (_, v) in zip(a.named_buffers(), b.named_buffers()):
            if name.endswith("num_batches_tracked"):
                assert int(u) == int(v), name
            else:
                np.testing.assert_allclose(u.numpy(), v.numpy(), rtol=1e-5, atol=1e-6, err_msg=name)

    # the reference's order
    ref = copy.deepcopy(h).train()
    run = RepresentationRunner(h, track_running_stats=True)
    with torch.no_grad():
        for i in range(G):
            ref(x1[i:i + 1])
    run(x1.cuda())
    run.sync_running_stats()
    train_step(h), train_step(ref)
    with torch.no_grad():
        for i in range(G):
            ref(x2[i:i + 1])
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        run(x2.cuda())          # notices the changed parameters, re-reads weights and buffers
        run.sync_running_stats()
    buffers_equal(h, ref)
    # no hand-over before training
    h2 = _net((3, 12, 12), (16, 3, 3), 2, [_cv(64, 2, 2, 0), _rb(64), _rb(3)], [_rb(3)])
    run2 = RepresentationRunner(h2, track_running_stats=True)
    run2(x1.cuda())
    train_step(h2)
    after_training = copy.deepcopy(h2)
    with pytest.warns(RuntimeWarning, match="dropped"):
        run2.refresh()
    for (name, u), (_, v) in zip(h2.named_buffers(), after_training.named_buffers()):
        if not name.endswith("num_batches_tracked"):
            assert torch.equal(u, v), name
    # and self-play goes on from the training step's values
    ref2 = copy.deepcopy(after_training).train()
    with torch.no_grad():
        for i in range(G):
            ref2(x2[i:i + 1])
    run2(x2.cuda())
    run2.sync_running_stats()
    for (name, u), (_, v) in zip(h2.named_buffers(), ref2.named_buffers()):
        if not name.endswith("num_batches_tracked"):
            np.testing.assert_allclose(u.numpy(), v.numpy(), rtol=1e-5, atol=1e-6, err_msg=name)


def _shipped_h():
    """The representation network of src/config/config_car_racing.yaml:57-131."""
    return _net((3, 96, 96), (256, 6, 6), 32,
                [_cv(128, 2, 2, 0), _rb(128), _rb(128), _cv(256, 2, 2, 0), _rb(256), _rb(256), _rb(256), _pool(2, 2, "max")],
                [_rb(3)], seed=3)


@pytest.mark.parametrize("fast", [True, False], ids=["split-fp16", "fp32-only"])
def test_shipped_representation_network_256_games(fast):
    """The shipped CarRacing h (32-frame history of 3x96x96 -> latent [256, 6, 6]) for 256 games in one
    call against the torch module called once per game on the GPU (float32, MIOpen)."""
    from muzero_amd.conv_repr import ConvRepresentation
    h = _shipped_h()
    G = 256
    gen = torch.Generator(device="cuda").manual_seed(5)
    x = torch.rand(G, 128, 96, 96, device="cuda", generator=gen) * 255.0  # CarRacing frames are 0..255 (car_racing.py:66)
    x[:, 96:] = torch.randint(0, 5, (G, 32, 1, 1), device="cuda", generator=gen).float()  # action planes
    run = ConvRepresentation(h, fast_convolutions=fast)
    got = run(x)
    torch.cuda.synchronize()
    want, _ = _per_game(h, x, torch.float32, device="cuda")
    assert got.shape == (G, 256, 6, 6)
    scale = float(want.abs().max())
    err = (got - want).abs().amax(dim=(1, 2, 3)) / scale
    print(f"shipped h, 256 games: max |HIP - torch| / scale = {float(err.max()):.2e} (median game {float(err.median()):.2e})")
    # float64 yardstick for three of the games (CPU): both float32 results sit at the same distance from it
    idx = [0, 100, 255]
    want64, _ = _per_game(h, x[idx].cpu(), torch.float64)
    ours = float((got[idx].double().cpu() - want64).abs().max()) / scale
    theirs = float((want[idx].double().cpu() - want64).abs().max()) / scale
    print(f"against float64: ours {ours:.2e}, float32 torch {theirs:.2e}")
    assert ours <= 1e-5 or ours <= 2 * theirs, (ours, theirs)
    assert float(err.max()) <= 1e-5 + 2 * (ours + theirs)


def test_staged_range_convolution_equals_the_gathering_kernel_bit_for_bit():
    """mz_conv3_same_forward against mz_conv3_forward on the same operands: both run the same products
    in the same order, so the outputs are identical bits -- at an image edge, across images, with a
    ragged last tile, for 3x3 and 1x1 kernels, float32 and split outputs."""
    import ctypes as C
    from muzero_amd import _hip
    from muzero_amd._hip import ptr
    lib = _hip.lib()
    gen = torch.Generator(device="cuda").manual_seed(11)
    for (G, H, W, Cin, N, k) in [(3, 7, 9, 64, 128, 3), (2, 24, 24, 128, 256, 3), (5, 6, 63, 64, 128, 3), (2, 5, 100, 64, 128, 1),
                                 (3, 5, 7, 192, 384, 3), (1, 1, 1, 64, 128, 3), (40, 3, 3, 64, 128, 3)]:
        rows, taps = G * H * W, k * k
        x = torch.randn(rows * Cin, device="cuda", generator=gen)
        w = torch.randn(N * taps * Cin, device="cuda", generator=gen) * 0.05
        bias = torch.randn(N, device="cuda", generator=gen)
        x3 = torch.empty((2, rows * Cin), dtype=torch.int16, device="cuda")
        w3 = torch.empty((2, N * taps * Cin), dtype=torch.int16, device="cuda")
        _hip.check(lib.mz_split3(ptr(x), x.numel(), ptr(x3), x3.shape[1], None), "mz_split3")
        _hip.check(lib.mz_split3(ptr(w), w.numel(), ptr(w3), w3.shape[1], None), "mz_split3")
        idx = torch.empty((taps, rows), dtype=torch.int32, device="cuda")
        _hip.check(lib.mz_conv_index(ptr(idx), G, H, W, k, 1, (k - 1) // 2, H, W, None), "mz_conv_index")
        outs = []
        for same in (False, True):
            y = torch.full((rows * N,), float("nan"), device="cuda")
            y3 = torch.zeros((2, rows * N), dtype=torch.int16, device="cuda")
            if same:
                _hip.check(lib.mz_conv3_same_forward(ptr(x3), x3.shape[1], Cin, G, H, W, k, ptr(w3), w3.shape[1], ptr(bias), 1,
                                                     ptr(y), ptr(y3), y3.shape[1], N, None, None), "mz_conv3_same_forward")
            else:
                _hip.check(lib.mz_conv3_forward(ptr(x3), x3.shape[1], Cin, ptr(idx), taps, ptr(w3), w3.shape[1], ptr(bias), 1,
                                                ptr(y), ptr(y3), y3.shape[1], rows, N, None, None), "mz_conv3_forward")
            outs.append((y, y3))
        torch.cuda.synchronize()
        assert torch.equal(outs[0][0], outs[1][0]), (G, H, W, Cin, N, k)
        assert torch.equal(outs[0][1], outs[1][1]), (G, H, W, Cin, N, k)
        # and the product is right: float64 convolution of the same operands
        ref = torch.nn.functional.conv2d(x.view(G, H, W, Cin).permute(0, 3, 1, 2).double().cpu(),
                                         w.view(N, k, k, Cin).permute(0, 3, 1, 2).double().cpu(), bias.double().cpu(),
                                         padding=(k - 1) // 2).relu().permute(0, 2, 3, 1).reshape(-1)
        _close(outs[1][0], ref)
    # an image too wide for the staged range is refused, not mis-computed
    assert lib.mz_conv3_same_forward(ptr(x3), x3.shape[1], 64, 1, 4, 64, 3, ptr(w3), w3.shape[1], None, 0, ptr(y), None, 0, 128, None, None) != 0
    assert b"width" in lib.mz_last_error()


def test_staged_range_on_wide_images_equals_the_gathering_kernel_bit_for_bit():
    """Images wider than 63 pixels (FlappyBird's layers are 144 and 72 wide, src/config/config_flappy_bird.yaml:60-99)
    are staged from the float32 form in 4 or 5 groups of slots: plain (the operand split while staged) against
    mz_split3 + mz_conv3_forward, and with the previous layer's per-game BatchNorm + activation applied while staging
    against mz_bn_apply3 + mz_conv3_forward -- identical bits, across image edges, games and a ragged last tile."""
    from muzero_amd import _hip
    from muzero_amd._hip import ptr
    lib = _hip.lib()
    gen = torch.Generator(device="cuda").manual_seed(13)
    for (G, H, W, Cin, N) in [(2, 5, 144, 128, 128), (1, 9, 72, 256, 256), (3, 4, 100, 64, 128), (2, 3, 64, 64, 128), (5, 2, 130, 64, 128)]:
        k, rows, taps, P = 3, G * H * W, 9, H * W
        assert lib.mz_conv3_same_fits(Cin, G, H, W, k, N, 1, 1) == 1 and lib.mz_conv3_same_fits(Cin, G, H, W, k, N, 0, 0) == 0
        x = torch.randn(rows * Cin, device="cuda", generator=gen)
        w = torch.randn(N * taps * Cin, device="cuda", generator=gen) * 0.05
        bias = torch.randn(N, device="cuda", generator=gen)
        scale = torch.rand(G, Cin, device="cuda", generator=gen) + 0.5
        shift = torch.randn(G, Cin, device="cuda", generator=gen)
        w3 = torch.empty((2, w.numel()), dtype=torch.int16, device="cuda")
        _hip.check(lib.mz_split3(ptr(w), w.numel(), ptr(w3), w3.shape[1], None), "mz_split3")
        idx = torch.empty((taps, rows), dtype=torch.int32, device="cuda")
        _hip.check(lib.mz_conv_index(ptr(idx), G, H, W, k, 1, 1, H, W, None), "mz_conv_index")
        for with_bn in (False, True):
            a3 = torch.empty((2, x.numel()), dtype=torch.int16, device="cuda")
            if with_bn:
                _hip.check(lib.mz_bn_apply3(ptr(x), ptr(scale), ptr(shift), None, None, None, None, 0, 1, None, ptr(a3), a3.shape[1],
                                            G, P, Cin, None), "mz_bn_apply3")
            else:
                _hip.check(lib.mz_split3(ptr(x), x.numel(), ptr(a3), a3.shape[1], None), "mz_split3")
            want = torch.full((rows * N,), float("nan"), device="cuda")
            want3 = torch.zeros((2, rows * N), dtype=torch.int16, device="cuda")
            _hip.check(lib.mz_conv3_forward(ptr(a3), a3.shape[1], Cin, ptr(idx), taps, ptr(w3), w3.shape[1], ptr(bias), 1,
                                            ptr(want), ptr(want3), want3.shape[1], rows, N, None, None), "mz_conv3_forward")
            got = torch.full((rows * N,), float("nan"), device="cuda")
            got3 = torch.zeros((2, rows * N), dtype=torch.int16, device="cuda")
            _hip.check(lib.mz_conv3_same_bn_forward(ptr(x), ptr(scale) if with_bn else None, ptr(shift) if with_bn else None, 1,
                                                    Cin, G, H, W, k, ptr(w3), w3.shape[1], ptr(bias), 1, ptr(got), ptr(got3),
                                                    got3.shape[1], N, None, None), "mz_conv3_same_bn_forward")
            torch.cuda.synchronize()
            assert torch.equal(got, want) and torch.equal(got3, want3), (G, H, W, Cin, N, with_bn)
    # beyond 144 pixels the range no longer fits: refused by the query and by the entry point
    assert lib.mz_conv3_same_fits(64, 1, 4, 200, 3, 128, 1, 0) == 0
    assert lib.mz_conv3_same_bn_forward(ptr(x), None, None, 0, 64, 1, 4, 200, 3, ptr(w3), w3.shape[1], None, 0, ptr(got), None, 0,
                                        128, None, None) != 0
    assert b"width" in lib.mz_last_error()


def _flappy_h():
    """The representation network of src/config/config_flappy_bird.yaml:58-165."""
    return _net((3, 512, 288), (256, 6, 6), 32,
                [_cv(128, 2, 2, 0), _rb(128), _rb(128), _cv(256, 2, 2, 0), _rb(256), _rb(256), _rb(256), _pool(2, 2, "avg"),
                 _rb(256), _rb(256), _rb(256), _pool(2, 2, "avg")], [_rb(3)], seed=4)


def test_shipped_flappy_representation_network():
    """FlappyBird's shipped h (32-frame history of 3x512x288 -> latent [256, 6, 6]) for 4 games in one call against
    the torch module called once per game on the GPU (float32, MIOpen); every ResBlock convolution of its 144-, 72-
    and 36-pixel-wide layers runs on the staged-range kernel."""
    from muzero_amd.conv_repr import ConvRepresentation
    h = _flappy_h()
    G = 4
    gen = torch.Generator(device="cuda").manual_seed(6)
    x = torch.rand(G, 128, 512, 288, device="cuda", generator=gen) * 255.0
    x[:, 96:] = torch.randint(0, 2, (G, 32, 1, 1), device="cuda", generator=gen).float()  # action planes
    run = ConvRepresentation(h)
    for (Gq, Hq, Wq, Cq) in [(G, 256, 144, 128), (G, 128, 72, 256), (G, 64, 36, 256)]:
        c = next(cv for cv in run.convs if cv.k == 3 and cv.cin == Cq)
        assert run._staged(c, Gq, Hq, Wq) and run._staged(c, Gq, Hq, Wq, with_bn=True)
    got = run(x)
    torch.cuda.synchronize()
    want, _ = _per_game(h, x, torch.float32, device="cuda")
    assert got.shape == (G, 256, 6, 6)
    scale = float(want.abs().max())
    err = (got - want).abs().amax(dim=(1, 2, 3)) / scale
    want64, _ = _per_game(h, x[:1].cpu(), torch.float64)
    ours = float((got[:1].double().cpu() - want64).abs().max()) / scale
    theirs = float((want[:1].double().cpu() - want64).abs().max()) / scale
    print(f"flappy h, {G} games: max |HIP - torch| / scale = {float(err.max()):.2e}; against float64: ours {ours:.2e}, float32 torch {theirs:.2e}")
    assert ours <= 1e-5 or ours <= 2 * theirs, (ours, theirs)
    assert float(err.max()) <= 1e-5 + 2 * (ours + theirs)


def test_block_sums_give_the_statistics_of_the_separate_pass():
    """The column sums a convolution's epilogue leaves per 64-row block, turned into per-game BatchNorm
    scale / shift / mean / variance by mz_bn_stats_from_sums, against mz_bn_stats on the same output
    (both add in float64; only the order of the additions differs) -- for both convolution kernels."""
    from muzero_amd import _hip
    from muzero_amd._hip import ptr
    lib = _hip.lib()
    gen = torch.Generator(device="cuda").manual_seed(12)
    G, H, W, Cin, N, k = 5, 8, 16, 64, 128, 3   # 128 pixels per game = 2 blocks; 640 rows = 2.5 tiles
    rows, taps, P = G * H * W, k * k, H * W
    x = torch.randn(rows * Cin, device="cuda", generator=gen) * 3.0 + 1.0
    w = torch.randn(N * taps * Cin, device="cuda", generator=gen) * 0.05
    bias = torch.randn(N, device="cuda", generator=gen)
    gamma = torch.rand(N, device="cuda", generator=gen) + 0.5
    beta = torch.randn(N, device="cuda", generator=gen)
    x3 = torch.empty((2, rows * Cin), dtype=torch.int16, device="cuda")
    w3 = torch.empty((2, N * taps * Cin), dtype=torch.int16, device="cuda")
    _hip.check(lib.mz_split3(ptr(x), x.numel(), ptr(x3), x3.shape[1], None), "mz_split3")
    _hip.check(lib.mz_split3(ptr(w), w.numel(), ptr(w3), w3.shape[1], None), "mz_split3")
    idx = torch.empty((taps, rows), dtype=torch.int32, device="cuda")
    _hip.check(lib.mz_conv_index(ptr(idx), G, H, W, k, 1, 1, H, W, None), "mz_conv_index")
    f32 = dict(dtype=torch.float32, device="cuda")
    for same in (False, True):
        y = torch.empty(rows * N, **f32)
        sums = torch.full((rows // 64, N, 2), float("nan"), dtype=torch.float64, device="cuda")
        if same:
            _hip.check(lib.mz_conv3_same_forward(ptr(x3), x3.shape[1], Cin, G, H, W, k, ptr(w3), w3.shape[1], ptr(bias), 0,
                                                 ptr(y), None, 0, N, ptr(sums), None), "mz_conv3_same_forward")
        else:
            _hip.check(lib.mz_conv3_forward(ptr(x3), x3.shape[1], Cin, ptr(idx), taps, ptr(w3), w3.shape[1], ptr(bias), 0,
                                            ptr(y), None, 0, rows, N, ptr(sums), None), "mz_conv3_forward")
        got = [torch.empty((G, N), **f32) for _ in range(4)]
        want = [torch.empty((G, N), **f32) for _ in range(4)]
        _hip.check(lib.mz_bn_stats_from_sums(ptr(sums), G, P, N, ptr(gamma), ptr(beta), 1e-5, *[ptr(t) for t in got], None),
                   "mz_bn_stats_from_sums")
        _hip.check(lib.mz_bn_stats(ptr(y), G, P, N, ptr(gamma), ptr(beta), 1e-5, *[ptr(t) for t in want], None), "mz_bn_stats")
        torch.cuda.synchronize()
        yb = y.view(rows // 64, 64, N).double()
        assert torch.allclose(sums[..., 0], yb.sum(1), rtol=1e-12, atol=1e-9)
        assert torch.allclose(sums[..., 1], (yb * yb).sum(1), rtol=1e-12, atol=1e-9)
        for a, b, name in zip(got, want, ("scale", "shift", "mean", "var")):
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-7, atol=1e-7, err_msg=name)
    # games that are not whole blocks are refused
    assert lib.mz_bn_stats_from_sums(ptr(sums), G, 100, N, ptr(gamma), ptr(beta), 1e-5, *[ptr(t) for t in got], None) != 0


def test_fused_forms_reproduce_the_separate_kernels_bit_for_bit():
    """Each fusion switch of ConvRepresentation changes where work is done, not what is computed:
    BatchNorm + activation applied while the next convolution stages its input, and the staged pixel
    range itself, give the bits of the separate kernels; statistics from block sums agree to the last
    float32 digit or two (float64 sums added in another order)."""
    from muzero_amd.conv_repr import ConvRepresentation
    h = _net((2, 16, 16), (256, 2, 2), 31, [_cv(128, 2, 2, 0), _rb(128), _rb(256, act="tanh"), _cv(256, 3, 1, 1, act="sigmoid"),
                                             _rb(256, k=3, s=2, p=1)], [])
    torch.manual_seed(4)
    x = torch.rand(5, 2 * 31 + 31, 16, 16).cuda()
    full = ConvRepresentation(h)(x).clone()
    no_bn = ConvRepresentation(h, fused_input_bn=False)(x).clone()
    no_range = ConvRepresentation(h, fused_input_bn=False, staged_range=False)(x).clone()
    no_sums = ConvRepresentation(h, fused_statistics=False)(x).clone()
    no_block = ConvRepresentation(h, fused_block_output=False)(x).clone()
    planes = ConvRepresentation(h, f32_gather=False)(x).clone()      # strided layers gathering from split planes
    planes_all = ConvRepresentation(h, f32_gather=False, staged_range=False, fused_input_bn=False)(x).clone()
    torch.cuda.synchronize()
    assert torch.equal(full, no_bn)
    assert torch.equal(no_bn, no_range)
    assert torch.equal(full, no_block)
    assert torch.equal(full, planes) and torch.equal(full, planes_all)
    np.testing.assert_allclose(full.cpu().numpy(), no_sums.cpu().numpy(), rtol=1e-6, atol=1e-6 * float(full.abs().max()))
    # a block's bn2 + skip sum + activation formed while the next block's conv1 stages its input: identity skips and
    # skips through a 1x1 convolution with its own BatchNorm, three blocks in a row, relu and tanh
    for acts in (("relu", "relu", "relu"), ("tanh", "relu", "sigmoid")):
        h2 = _net((3, 12, 12), (128, 3, 3), 5, [_cv(64, 2, 2, 0), _rb(128, act=acts[0]), _rb(128, act=acts[1]), _rb(256, act=acts[2]),
                                                _rb(256, act=acts[0])], [])
        x2 = torch.rand(7, 3 * 5 + 5, 12, 12).cuda()
        run = ConvRepresentation(h2)
        fused = run(x2).clone()
        plain = ConvRepresentation(h2, fused_block_output=False)(x2).clone()
        torch.cuda.synchronize()
        assert torch.equal(fused, plain), acts
        want64, _ = _per_game(h2, x2.cpu(), torch.float64)
        _close(fused, want64)


def test_staged_convolution_applies_the_previous_batchnorm_bit_for_bit():
    """mz_conv3_same_bn_forward(x, scale, shift, relu) against mz_bn_apply3 (to split planes) followed by
    mz_conv3_same_forward: identical bits, for ranges that span several small games and for a NULL scale
    (the float32 input split on the fly) against mz_split3 + mz_conv3_same_forward."""
    from muzero_amd import _hip
    from muzero_amd._hip import ptr
    lib = _hip.lib()
    gen = torch.Generator(device="cuda").manual_seed(13)
    for (G, H, W, Cin, N) in [(4, 24, 24, 128, 128), (37, 4, 5, 64, 256), (3, 9, 8, 192, 128)]:
        rows, P, k = G * H * W, H * W, 3
        x = torch.randn(rows * Cin, device="cuda", generator=gen) * 2.0
        scale = torch.rand((G, Cin), device="cuda", generator=gen) + 0.5
        shift = torch.randn((G, Cin), device="cuda", generator=gen)
        w = torch.randn(N * 9 * Cin, device="cuda", generator=gen) * 0.05
        w3 = torch.empty((2, w.numel()), dtype=torch.int16, device="cuda")
        _hip.check(lib.mz_split3(ptr(w), w.numel(), ptr(w3), w3.shape[1], None), "mz_split3")
        a3 = torch.empty((2, rows * Cin), dtype=torch.int16, device="cuda")
        _hip.check(lib.mz_bn_apply3(ptr(x), ptr(scale), ptr(shift), None, None, None, None, 0, 1, None, ptr(a3), a3.shape[1], G, P, Cin,
                                    None), "mz_bn_apply3")
        want, got = torch.empty(rows * N, device="cuda"), torch.empty(rows * N, device="cuda")
        _hip.check(lib.mz_conv3_same_forward(ptr(a3), a3.shape[1], Cin, G, H, W, k, ptr(w3), w3.shape[1], None, 0, ptr(want), None, 0, N,
                                             None, None), "mz_conv3_same_forward")
        _hip.check(lib.mz_conv3_same_bn_forward(ptr(x), ptr(scale), ptr(shift), 1, Cin, G, H, W, k, ptr(w3), w3.shape[1], None, 0,
                                                ptr(got), None, 0, N, None, None), "mz_conv3_same_bn_forward")
        torch.cuda.synchronize()
        assert torch.equal(got, want), (G, H, W, Cin, N)
        x3 = torch.empty((2, rows * Cin), dtype=torch.int16, device="cuda")
        _hip.check(lib.mz_split3(ptr(x), x.numel(), ptr(x3), x3.shape[1], None), "mz_split3")
        _hip.check(lib.mz_conv3_same_forward(ptr(x3), x3.shape[1], Cin, G, H, W, k, ptr(w3), w3.shape[1], None, 0, ptr(want), None, 0, N,
                                             None, None), "mz_conv3_same_forward")
        _hip.check(lib.mz_conv3_same_bn_forward(ptr(x), None, None, 0, Cin, G, H, W, k, ptr(w3), w3.shape[1], None, 0, ptr(got), None, 0,
                                                N, None, None), "mz_conv3_same_bn_forward")
        torch.cuda.synchronize()
        assert torch.equal(got, want), (G, H, W, Cin, N)


def test_operands_beyond_the_fp16_range_are_loud():
    """The split-product kernels hold an operand as two fp16 terms: a value beyond 65504 has no such form.  The result
    is then non-finite -- visible -- and never a wrong finite number; values just inside the range are exact to the
    usual tolerance."""
    from muzero_amd import _hip
    from muzero_amd._hip import ptr
    lib = _hip.lib()
    G, H, W, Cin, N = 1, 8, 8, 64, 128
    rows = G * H * W
    gen = torch.Generator(device="cuda").manual_seed(21)
    w = torch.randn(N * 9 * Cin, device="cuda", generator=gen) * 0.05
    w3 = torch.empty((2, w.numel()), dtype=torch.int16, device="cuda")
    _hip.check(lib.mz_split3(ptr(w), w.numel(), ptr(w3), w3.shape[1], None), "mz_split3")
    for peak, finite in ((6.0e4, True), (1.0e5, False)):
        x = torch.rand(rows * Cin, device="cuda", generator=gen)
        x[5 * Cin + 3] = peak
        y = torch.empty(rows * N, device="cuda")
        _hip.check(lib.mz_conv3_same_bn_forward(ptr(x), None, None, 0, Cin, G, H, W, 3, ptr(w3), w3.shape[1], None, 0, ptr(y), None, 0,
                                                N, None, None), "mz_conv3_same_bn_forward")
        torch.cuda.synchronize()
        assert bool(torch.isfinite(y).all()) == finite
        if finite:
            ref = torch.nn.functional.conv2d(x.view(G, H, W, Cin).permute(0, 3, 1, 2).double().cpu(),
                                             w.view(N, 3, 3, Cin).permute(0, 3, 1, 2).double().cpu(), padding=1).permute(0, 2, 3, 1).reshape(-1)
            _close(y, ref)
