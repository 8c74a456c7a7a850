"""GPU parity of the fused fp32 MFMA network kernels against torch fp32 on the CPU and the
reference-generated c4_mlp vectors.  Tolerance: 1e-5 relative (the contract's bound for
network outputs; differences come from summation order only)."""
import os

import numpy as np
import pytest
import torch

from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-5, 2e-6


def _c4_nets(seed=0):
    from muzero_amd.config import load_config
    from muzero_amd.neural_networks import DynamicsNetwork, PredictionNetwork, RepresentationNetwork
    cfg = load_config("config_c4_mlp.yaml")
    torch.manual_seed(seed)
    h = RepresentationNetwork((2, 6, 7), cfg.networks.latent_shape, cfg.networks.representation)
    g = DynamicsNetwork(cfg.networks.latent_shape, 7, cfg.networks.dynamics)
    f = PredictionNetwork(cfg.networks.latent_shape, 7, cfg.networks.prediction)
    return h, g, f


def _custom_nets(latent, A, reward, value, policy, acts=("relu",), seed=3):
    from muzero_amd.config import DenseLayerConfig, DynamicsNetworkConfig, PredictionNetworkConfig
    from muzero_amd.neural_networks import DynamicsNetwork, PredictionNetwork
    torch.manual_seed(seed)
    mk = lambda ws: [DenseLayerConfig(out_features=n, activation=acts[i % len(acts)]) for i, n in enumerate(ws)]  # noqa: E731
    g = DynamicsNetwork(latent, A, DynamicsNetworkConfig(res_net=[], reward_net=mk(reward)))
    f = PredictionNetwork(latent, A, PredictionNetworkConfig(res_net=[], value_net=mk(value), policy_net=mk(policy)))
    return g, f


@pytest.mark.parametrize("B", [1, 7, 16, 33, 4096, 28672])
def test_c4_mlp_networks_match_torch(B):
    from muzero_amd.engine import PackedNetworks
    _, g, f = _c4_nets()
    nets = PackedNetworks(g, f)
    torch.manual_seed(B)
    lat = torch.randn(B, 4, 6, 7)
    act = torch.randint(0, 7, (B,))
    with torch.no_grad():
        want_next, want_r = g(lat, act)
        want_p, want_v = f(lat)
        want_prior = torch.softmax(want_p, dim=1)
    d_lat = lat.reshape(B, -1).cuda().contiguous()
    nxt = torch.empty(B, 168, device="cuda")
    r = torch.empty(B, device="cuda")
    nets.dynamics(d_lat, B, action=act.int().cuda(), next_latent=nxt, reward=r)
    p = torch.empty(B, 7, device="cuda")
    pri = torch.empty(B, 7, device="cuda")
    v = torch.empty(B, device="cuda")
    am = torch.empty(B, dtype=torch.int32, device="cuda")
    nets.prediction(d_lat, B, policy=p, value=v, argmax=am)
    nets.prediction(d_lat, B, softmax_times=1, policy=pri)
    torch.cuda.synchronize()
    np.testing.assert_allclose(nxt.cpu().numpy(), want_next.reshape(B, -1).numpy(), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(r.cpu().numpy(), want_r[:, 0].numpy(), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(p.cpu().numpy(), want_p.numpy(), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(pri.cpu().numpy(), want_prior.numpy(), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(v.cpu().numpy(), want_v[:, 0].numpy(), rtol=RTOL, atol=ATOL)
    # argmax must agree wherever the top two probabilities are not within rounding of each other
    top2 = want_p.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-6
    assert torch.equal(am.cpu()[clear].long(), want_p.argmax(dim=1)[clear])


def test_expansion_pattern_rows_and_scatter():
    """One latent per tree evaluated for all A actions (action = row % A), children written
    straight into a [G, cap, D] pool through the row scatter."""
    from muzero_amd.engine import PackedNetworks
    _, g, f = _c4_nets(1)
    nets = PackedNetworks(g, f)
    G, A, D, cap = 37, 7, 168, 15
    torch.manual_seed(5)
    pool = torch.zeros(G, cap, D)
    pool[:, 3] = torch.randn(G, D)
    d_pool = pool.cuda()
    rows = (torch.arange(G, dtype=torch.int32) * cap + 3).repeat_interleave(A).cuda()
    next_rows = ((torch.arange(G, dtype=torch.int32) * cap + 8)[:, None] + torch.arange(A, dtype=torch.int32)).reshape(-1).cuda()
    rew = torch.empty(G * A, device="cuda")
    nets.dynamics(d_pool, G * A, rows=rows, action=None, next_latent=d_pool, next_rows=next_rows, reward=rew)
    torch.cuda.synchronize()
    with torch.no_grad():
        want_next, want_r = g(pool[:, 3].reshape(G, 4, 6, 7).repeat_interleave(A, 0), torch.arange(A).repeat(G))
    got = d_pool.cpu()
    np.testing.assert_allclose(got[:, 8:15].reshape(G * A, D).numpy(), want_next.reshape(G * A, D).numpy(), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(rew.cpu().numpy(), want_r[:, 0].numpy(), rtol=RTOL, atol=ATOL)
    assert torch.equal(got[:, :8], pool[:, :8])  # nothing else touched


@pytest.mark.parametrize("latent,A,reward,value,policy,acts", [
    ((2, 3, 3), 5, [16], [16], [16], ("relu",)),
    ((1, 1, 8), 4, [20, 12], [33], [17, 9, 40], ("tanh", "sigmoid", "relu")),
    ((3, 5, 5), 2, [], [300], [64], ("relu",)),
    ((2, 4, 4), 18, [8], [8], [24], ("linear", "relu")),
])
def test_other_mlp_shapes(latent, A, reward, value, policy, acts):
    from muzero_amd.engine import PackedNetworks
    g, f = _custom_nets(latent, A, reward, value, policy, acts)
    nets = PackedNetworks(g, f)
    B, D = 50, latent[0] * latent[1] * latent[2]
    torch.manual_seed(9)
    lat = torch.randn(B, *latent)
    act = torch.randint(0, A, (B,))
    with torch.no_grad():
        want_next, want_r = g(lat, act)
        want_p, want_v = f(lat)
    d_lat = lat.reshape(B, -1).cuda().contiguous()
    nxt, r = torch.empty(B, D, device="cuda"), torch.empty(B, device="cuda")
    p, v = torch.empty(B, A, device="cuda"), torch.empty(B, device="cuda")
    nets.dynamics(d_lat, B, action=act.int().cuda(), next_latent=nxt, reward=r)
    nets.prediction(d_lat, B, policy=p, value=v)
    torch.cuda.synchronize()
    np.testing.assert_allclose(nxt.cpu().numpy(), want_next.reshape(B, -1).numpy(), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(r.cpu().numpy(), want_r[:, 0].numpy(), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(p.cpu().numpy(), want_p.numpy(), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(v.cpu().numpy(), want_v[:, 0].numpy(), rtol=RTOL, atol=ATOL)


def test_reference_vectors_c4_mlp():
    """Inputs/outputs captured from the reference's own modules (same seed-0 weights)."""
    from muzero_amd.engine import PackedNetworks
    z = np.load(os.path.join(GOLDEN, "c4_mlp.npz"))
    _, g, f = _c4_nets(0)
    nets = PackedNetworks(g, f)
    lat = torch.from_numpy(z["g_latent"])
    B = lat.shape[0]
    d_lat = lat.reshape(B, -1).cuda().contiguous()
    nxt, r = torch.empty(B, 168, device="cuda"), torch.empty(B, device="cuda")
    p, v = torch.empty(B, 7, device="cuda"), torch.empty(B, device="cuda")
    nets.dynamics(d_lat, B, action=torch.from_numpy(z["g_action"]).int().cuda(), next_latent=nxt, reward=r)
    nets.prediction(d_lat, B, policy=p, value=v)
    torch.cuda.synchronize()
    np.testing.assert_allclose(nxt.cpu().numpy(), z["g_next"].reshape(B, -1), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(r.cpu().numpy(), z["g_reward"][:, 0], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(p.cpu().numpy(), z["f_policy"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(v.cpu().numpy(), z["f_value"][:, 0], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("B0,B1", [(1301, 77), (8200, 1000), (500, 8195), (8192 + 17, 8192 + 40)],
                         ids=["16+16", "32+16", "16+32", "32+32"])
def test_two_heads_in_one_launch_equal_two_launches(B0, B1):
    """mz_mlp_forward_pair (reward head over gathered rows + value head over other rows, one launch) against two
    mz_mlp_forward calls: identical bits, ragged row counts, a strided one-output destination; each job with the row
    tile (16 or 32 rows per workgroup) its own batch size selects."""
    import ctypes as C
    from muzero_amd import _hip
    from muzero_amd._hip import ptr
    from muzero_amd.engine import PackedNetworks
    _, g, f = _c4_nets(1)
    nets = PackedNetworks(g, f)
    lib = _hip.lib()
    st = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator(device="cuda").manual_seed(4)
    D = nets.plan.D
    x0 = torch.randn(B0 + 5, D, device="cuda", generator=gen)
    r0 = torch.randperm(B0 + 5, device="cuda", generator=gen)[:B0].to(torch.int32)
    r0[3] = -1   # an absent row reads as zeros
    x1 = torch.randn(B1, D, device="cuda", generator=gen)
    p0, p1 = nets.plan.reward, nets.plan.value
    blob = nets.blob
    want0, want1 = torch.zeros(B0, device="cuda"), torch.zeros(B1, 2, device="cuda")
    _hip.check(lib.mz_mlp_forward(C.byref(p0), ptr(blob), ptr(x0), D, ptr(r0), ptr(want0), 1, B0, st), "a")
    _hip.check(lib.mz_mlp_forward(C.byref(p1), ptr(blob), ptr(x1), D, None, ptr(want1), 2, B1, st), "b")
    got0, got1 = torch.zeros(B0, device="cuda"), torch.zeros(B1, 2, device="cuda")
    _hip.check(lib.mz_mlp_forward_pair(C.byref(p0), ptr(x0), D, ptr(r0), ptr(got0), 1, B0,
                                       C.byref(p1), ptr(x1), D, None, ptr(got1), 2, B1, ptr(blob), st), "pair")
    torch.cuda.synchronize()
    assert torch.equal(got0, want0) and torch.equal(got1, want1)
    assert float(got1[:, 1].abs().max()) == 0.0 and float(got0.abs().max()) > 0.0
