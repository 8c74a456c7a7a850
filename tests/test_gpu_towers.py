"""Residual-tower g / f on the GPU through the C ABI:
  * the wide dense layer against torch fp32 (1e-5), every tiling / split / tail case;
  * the tower kernel and the composed g / f against the vectors captured from the reference
    (tests/golden/towers.npz): per-latent calls and the expansion call whose batch
    statistics couple the A children;
  * the lock-step search with these networks: its own network outputs replayed through the
    CPU tree oracle reproduce every tree bit for bit, and the reference's trees are
    reproduced whenever no near-tie flips.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import pyref, tree as otree
from tests.test_towers_golden import NETS, TREES, Z, load_tower_nets

pytestmark = pytest.mark.gpu

TOL = dict(rtol=1e-5, atol=2e-6)


def _wide(lin_w, lin_b, act, x, x_rows=None, ldw=0, K=None):
    from muzero_amd import _hip
    from muzero_amd._hip import ptr
    lib = _hip.lib()
    N = lin_w.shape[0]
    K = K or lin_w.shape[1]
    w = lin_w.cuda().contiguous()
    b = lin_b.cuda().contiguous() if lin_b is not None else None
    d = _hip.Dense(ptr(w), ptr(b), K, N, _hip.ACT[act], ldw)
    pl = _hip.PackedLayer()
    n = lib.mz_wide_plan(C.byref(d), 0, C.byref(pl))
    assert n > 0
    blob = torch.zeros(n, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _hip.check(lib.mz_wide_pack(C.byref(d), C.byref(pl), ptr(blob), st), "pack")
    xg = x.cuda().contiguous()
    rows = None if x_rows is None else torch.as_tensor(x_rows, dtype=torch.int32).cuda()
    B = xg.shape[0] if rows is None else rows.numel()
    need = lib.mz_wide_workspace(C.byref(pl), B)
    ws = torch.empty(max(need, 1), device="cuda")
    y = torch.full((B, N + 3), 7.0, device="cuda")  # row stride wider than N: padding must stay untouched
    _hip.check(lib.mz_wide_forward(C.byref(pl), ptr(blob), ptr(xg), xg.shape[1], ptr(rows), ptr(y), N + 3, B,
                                   ptr(ws), need, st), "forward")
    torch.cuda.synchronize()
    assert (y[:, N:] == 7.0).all()
    return y[:, :N].cpu(), need


@pytest.mark.parametrize("B,K,N,act", [
    (5, 36, 10, "none"),        # one small tile, K not a multiple of 16, odd N
    (64, 64, 64, "relu"),       # exactly one small tile
    (70, 130, 75, "tanh"),      # big tiles with tails everywhere, K % 4 != 0 -> scalar loads
    (256, 1024, 512, "relu"),   # big tiles, split reduction
    (300, 520, 130, "sigmoid"),
    (2, 9216, 24, "none"),      # the shipped head shape, tiny batch: deep split
    (1792, 256, 128, "relu"),
])
def test_wide_layer_matches_torch(B, K, N, act):
    torch.manual_seed(B + K + N)
    w = torch.randn(N, K) / K ** 0.5
    b = torch.randn(N)
    x = torch.randn(B, K)
    got, need = _wide(w, b, act, x)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    ref = {"none": lambda v: v, "relu": torch.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid}[act](ref).float()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)


def test_wide_layer_gather_submatrix_and_determinism():
    """Row gather with repeated and negative (= zero) rows; a layer that reads the left half of a
    wider weight matrix (fc_merge's latent half); two runs give identical bits."""
    torch.manual_seed(3)
    D = 200
    w = torch.randn(D, 2 * D) / D ** 0.5
    x = torch.randn(40, D)
    rows = [3, 3, -1, 39, 0, 17] * 30
    got, need = _wide(w, None, "none", x, x_rows=rows, ldw=2 * D, K=D)
    again, _ = _wide(w, None, "none", x, x_rows=rows, ldw=2 * D, K=D)
    assert need > 0  # this shape splits the reduction: the partial sums are combined in a fixed order
    assert torch.equal(got, again)
    xr = torch.stack([x[r] if r >= 0 else torch.zeros(D) for r in rows])
    ref = (xr.double() @ w[:, :D].double().t()).float()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("wide_above", [4096, 0], ids=["mlp-heads", "wide-heads"])
@pytest.mark.parametrize("ni", NETS)
def test_g_and_f_match_reference_vectors(ni, wide_above):
    from muzero_amd.towers import TowerNetworks
    g, f, A, shape = load_tower_nets(ni)
    D = int(np.prod(shape))
    nets = TowerNetworks(g, f, wide_above=wide_above)
    lat = torch.from_numpy(Z[f"net{ni}_latent"]).cuda().reshape(-1, D).contiguous()
    act = torch.from_numpy(Z[f"net{ni}_action"]).to(torch.int32).cuda()
    B = lat.shape[0]
    # --- B independent calls of batch size 1
    nxt = torch.zeros(B, D, device="cuda")
    rew = torch.zeros(B, device="cuda")
    nets.dynamics(lat, B, action=act, next_latent=nxt, reward=rew)
    pol = torch.zeros(B, A, device="cuda")
    pol2 = torch.zeros(B, A, device="cuda")
    val = torch.zeros(B, device="cuda")
    arg = torch.zeros(B, dtype=torch.int32, device="cuda")
    nets.prediction(lat, B, policy=pol, value=val, argmax=arg)
    nets.prediction(lat, B, softmax_times=1, policy=pol2)
    torch.cuda.synchronize()
    np.testing.assert_allclose(nxt.cpu().numpy(), Z[f"net{ni}_g1_next"].reshape(B, D), **TOL)
    np.testing.assert_allclose(rew.cpu().numpy(), Z[f"net{ni}_g1_reward"][:, 0], **TOL)
    np.testing.assert_allclose(pol.cpu().numpy(), Z[f"net{ni}_f1_policy"], **TOL)
    np.testing.assert_allclose(val.cpu().numpy(), Z[f"net{ni}_f1_value"][:, 0], **TOL)
    assert np.array_equal(arg.cpu().numpy(), Z[f"net{ni}_f1_policy"].argmax(1))
    want2 = torch.softmax(torch.from_numpy(Z[f"net{ni}_f1_policy"]), dim=1).numpy()
    np.testing.assert_allclose(pol2.cpu().numpy(), want2, **TOL)
    # --- expansion: every latent with all A actions as ONE call (statistics over the A children),
    # gathered from / scattered to rows of a larger buffer as the search does
    store = torch.zeros(B * (A + 1) + 5, D, device="cuda")
    src_rows = torch.arange(B, dtype=torch.int32, device="cuda") * (A + 1) + 2
    store[src_rows.long()] = lat
    rows_rep = src_rows.repeat_interleave(A)
    dst_rows = (src_rows[:, None] + 1 + torch.arange(A, dtype=torch.int32, device="cuda")[None]).reshape(-1).contiguous()
    rewA = torch.zeros(B, A, device="cuda")
    nets.dynamics(store, B * A, rows=rows_rep, action=None, next_latent=store, next_rows=dst_rows, reward=rewA)
    torch.cuda.synchronize()
    got = store[dst_rows.long()].reshape(B, A, D).cpu().numpy()
    np.testing.assert_allclose(got, Z[f"net{ni}_gA_next"].reshape(B, A, D), **TOL)
    np.testing.assert_allclose(rewA.cpu().numpy(), Z[f"net{ni}_gA_reward"][:, :, 0], **TOL)
    assert torch.equal(store[src_rows.long()], lat)  # sources untouched


def test_tower_rejects_what_it_cannot_hold():
    from muzero_amd import _hip
    from muzero_amd.towers import TowerNetworks
    g, f, A, shape = load_tower_nets(1)
    nets = TowerNetworks(g, f)
    lib = _hip.lib()
    x = torch.zeros(8, nets.D, device="cuda")
    # B not a multiple of the group
    rc = lib.mz_tower_forward(C.byref(nets.tower_g), x.data_ptr(), nets.D, None, None, 0, None, x.data_ptr(), nets.D,
                              None, None, 0, 7, 3, None)
    assert rc < 0 and b"multiple" in lib.mz_last_error()
    # one value per channel: torch raises for it too (BatchNorm2d in train mode)
    t1 = _hip.Tower.from_buffer_copy(nets.tower_g)
    t1.H = t1.W = 1
    rc = lib.mz_tower_forward(C.byref(t1), x.data_ptr(), nets.D, None, None, 0, None, x.data_ptr(), nets.D, None, None, 0, 4, 1, None)
    assert rc < 0 and b"more than 1 value" in lib.mz_last_error()


@pytest.mark.parametrize("ni,ti", TREES)
def test_lockstep_search_with_towers(ni, ti):
    """G copies of the reference's recorded search plus perturbed roots, run in lock step."""
    from muzero_amd.engine import BatchedMCTS
    from muzero_amd.towers import TowerNetworks
    k = f"net{ni}_t{ti}_"
    g, f, A, shape = load_tower_nets(ni)
    D = int(np.prod(shape))
    S, depth = int(Z[k + "S"]), int(Z[k + "depth"])
    noise, disc, tp = float(Z[k + "noise_frac"]), float(Z[k + "discount"]), int(Z[k + "to_play"])
    G = 6
    rng = np.random.default_rng(ni * 10 + ti)
    lat = np.repeat(Z[k + "root_latent"].reshape(1, D), G, 0).astype(np.float32)
    lat[2:] += rng.standard_normal((G - 2, D)).astype(np.float32) * 0.5
    eta = np.repeat(Z[k + "eta"][None], G, 0).astype(np.float32)
    rand = np.repeat(Z[k + "rand"][None], G, 0).astype(np.int32)
    rand[3:] = rng.integers(0, A, size=(G - 3, S + 1))
    to_play = np.full(G, tp, np.int8)
    nets = TowerNetworks(g, f)
    mcts = BatchedMCTS(nets, G, S, depth, discount=disc, noise_frac=noise)
    trace = {}
    res = mcts.run(torch.from_numpy(lat).cuda(), torch.from_numpy(to_play).cuda(), torch.from_numpy(eta).cuda(),
                   torch.from_numpy(rand).cuda(), trace=trace)
    torch.cuda.synchronize()
    pool = mcts.pool
    # --- the GPU's own network outputs replayed through the tree oracle: bit-exact trees
    prior_s = np.stack([p.cpu().numpy() for p in trace["prior"]], axis=1)
    rollout = np.stack([r.cpu().numpy() for r in trace["rollout"]], axis=1)
    reward_nodes = pool.reward.cpu().numpy()
    root_prior = torch.empty(G, A, device="cuda")
    nets.prediction(torch.from_numpy(lat).cuda(), G, softmax_times=1, policy=root_prior)
    prior_stream = np.concatenate([root_prior.cpu().numpy()[:, None], prior_s], axis=1)
    reward_stream = np.stack([reward_nodes[:, 1 + s * A:1 + (s + 1) * A] for s in range(S + 1)], axis=1)
    ref = otree.run_batch(A, S, depth, to_play, prior_stream, reward_stream, rand, rollout, eta, noise, disc)
    assert np.array_equal(np.stack([x.cpu().numpy() for x in trace["leaf"]], 1), ref["leaf"])
    assert np.array_equal(pool.visit.cpu().numpy(), ref["visit"])
    assert np.array_equal(pool.value_sum.cpu().numpy(), ref["value_sum"])
    assert np.array_equal(res.value.cpu().numpy(), ref["utility"])
    # --- trees 0 and 1 are the reference's recorded search: its network streams to 1e-5 wherever
    # the same nodes were visited, the whole tree when no near-tie flipped
    for gi in (0, 1):
        np.testing.assert_allclose(prior_stream[gi, 0], Z[k + "prior_stream"][0], **TOL)
        np.testing.assert_allclose(reward_stream[gi, 0], Z[k + "reward_stream"][0], **TOL)
        if np.array_equal(ref["leaf"][gi], Z[k + "leaf"]) and np.array_equal(ref["child"][gi], Z[k + "child"]):
            np.testing.assert_allclose(prior_stream[gi], Z[k + "prior_stream"], **TOL)
            np.testing.assert_allclose(reward_stream[gi], Z[k + "reward_stream"], **TOL)
            np.testing.assert_allclose(rollout[gi], Z[k + "sim_stream"], **TOL)
            assert np.array_equal(ref["visit"][gi], Z[k + "visit"])
            # a node's value_sum adds one backed-up utility per visit, signs mixed: the terms are
            # within 1e-5 of the reference's, the sum within visits * 1e-5 * (largest term)
            want = Z[k + "value_sum"]
            term = max(float(np.abs(Z[k + "sim_stream"]).max()), float(np.abs(Z[k + "reward_stream"]).max()))
            bound = 1e-5 * np.abs(want) + 1e-5 * term * np.maximum(Z[k + "visit"], 1)
            assert (np.abs(ref["value_sum"][gi] - want) <= bound).all()
            np.testing.assert_allclose(res.policy[gi].cpu().numpy(), Z[k + "policy"], rtol=1e-5)
    same_as_reference = sum(int(np.array_equal(ref["visit"][gi], Z[k + "visit"])) for gi in (0, 1))
    assert same_as_reference == 2, "the reference's recorded tree was not reproduced"


@pytest.mark.parametrize("ni,ti", TREES)
def test_dropin_mcts_run_with_towers_reproduces_reference_run(ni, ti):
    """create_mcts(...).run(root) -- the reference's call -- with residual-tower modules and the
    global RNGs seeded as the reference's run was."""
    import random
    from muzero_amd.config import MCTSConfig
    from muzero_amd.search.factory import create_mcts
    from muzero_amd.search.nodes import Node
    k = f"net{ni}_t{ti}_"
    g, f, A, shape = load_tower_nets(ni)
    S, depth = int(Z[k + "S"]), int(Z[k + "depth"])
    cfg = MCTSConfig(selection_strategy="puct", max_iterations=S, max_time=0, model_look_ahead=depth,
                     discount_factor=float(Z[k + "discount"]), dirichlet_alpha=0.3, noise_frac=float(Z[k + "noise_frac"]))
    mcts = create_mcts(g, f, torch.arange(A).long(), cfg)
    seed = int(Z[k + "seed"])
    random.seed(seed)
    torch.manual_seed(seed)
    root = Node(latent_state=torch.from_numpy(Z[k + "root_latent"]), to_play=int(Z[k + "to_play"]))
    policy, utility = mcts.run(root)
    kids = list(root.children.values())
    assert root.visit_count == S and len(kids) == A
    np.testing.assert_allclose([float(c.policy_priority) for c in kids], Z[k + "prior"][1:1 + A], rtol=1e-5)
    assert [c.visit_count for c in kids] == Z[k + "visit"][1:1 + A].tolist()
    np.testing.assert_allclose(policy, Z[k + "policy"], rtol=1e-5)
    np.testing.assert_allclose(utility, float(Z[k + "utility"]), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose([c.value_sum for c in kids], Z[k + "value_sum"][1:1 + A], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("ni", NETS)
def test_dropin_expand_and_simulate_with_towers(ni):
    """expand_node and MuZeroSimulation on Node objects (the reference's unit-test surface):
    the children of one expansion are normalised together, a rollout step alone."""
    from muzero_amd.search.expansion import expand_node
    from muzero_amd.search.nodes import Node
    from muzero_amd.search.simulation import MuZeroSimulation
    g, f, A, shape = load_tower_nets(ni)
    lat = torch.from_numpy(Z[f"net{ni}_latent"])
    for i in range(2):
        node = Node(latent_state=lat[i:i + 1].clone(), to_play=1)
        expand_node(node, torch.arange(A).long(), g, f)
        kids = list(node.children.values())
        got = torch.cat([c.latent_state for c in kids]).numpy()
        np.testing.assert_allclose(got, Z[f"net{ni}_gA_next"][i], **TOL)
        np.testing.assert_allclose([float(c.reward) for c in kids], Z[f"net{ni}_gA_reward"][i][:, 0], **TOL)
        want_prior = torch.softmax(torch.from_numpy(Z[f"net{ni}_f1_policy"][i:i + 1]), dim=1)[0].numpy()
        np.testing.assert_allclose([c.policy_priority for c in kids], want_prior, **TOL)
    # one greedy rollout step: f's argmax, g alone, f's value -- against the torch modules
    node = Node(latent_state=lat[3:4].clone(), to_play=1)
    out = MuZeroSimulation(g, f, depth=1)(node)
    with torch.no_grad():
        a = f(lat[3:4])[0].argmax(dim=1)
        nl, r = g(lat[3:4], a)
        v = f(nl)[1]
    np.testing.assert_allclose(out, [float(r), float(v)], rtol=1e-5, atol=2e-6)


def test_selfplay_with_host_environments_conv_h_and_towers():
    """Configs 3-4 in small: single-player simulators on the host (to_play 0), a convolutional h with
    BatchNorm evaluated with per-game statistics, residual-tower g / f in the lock-step search.
    The first move of every game is recomputed by hand: h one game at a time on the CPU, then the
    drop-in MCTS.run fed with the engine's own random streams."""
    from muzero_amd.config import (ConvLayerConfig, MCTSConfig, RepresentationNetworkConfig, ResBlockConfig,
                                   TrainingDataGeneratorConfig)
    from muzero_amd.engine import BatchedMCTS
    from muzero_amd.environments.host_pool import HostEnvPool
    from muzero_amd.neural_networks import RepresentationNetwork
    from muzero_amd.selfplay import SelfPlayEngine
    from muzero_amd.towers import TowerNetworks
    from tests.test_host_pool import CountingEnv
    g, f, A, shape = load_tower_nets(0)          # latent (16, 4, 4), 5 actions
    torch.manual_seed(4)
    rb = ResBlockConfig(out_channels=6, kernel_size=3, stride=1, padding=1, activation="relu", pool_kernel_size=1,
                        pool_stride=1)
    K = 2
    h = RepresentationNetwork((2, 8, 8), shape, RepresentationNetworkConfig(history_length=K, downsample=[
        ConvLayerConfig(out_channels=6, kernel_size=2, stride=2, padding=0, activation="relu"), rb], res_net=[]))

    class Env(CountingEnv):
        def get_to_play(self):
            return 0

        def _obs(self):
            gen = torch.Generator().manual_seed(self.gid * 1000 + self.t)
            return torch.rand((1,) + self.shape, generator=gen)

    G, S = 6, 12
    envs = [Env(gi, 3 + gi % 3, shape=(2, 8, 8), actions=A) for gi in range(G)]
    cfg = TrainingDataGeneratorConfig(
        num_episodes=1, max_steps_per_episode=50, total_time=0, epsilon=0.0, epsilon_decay=1.0, min_epsilon=0.0,
        mcts=MCTSConfig(selection_strategy="puct", max_iterations=S, max_time=0, model_look_ahead=1,
                        discount_factor=0.998, dirichlet_alpha=0.3, noise_frac=0.25))
    eng = SelfPlayEngine(h, g, f, G, cfg, seed=3, env=HostEnvPool(envs, workers=2))
    assert isinstance(eng.nets, TowerNetworks)
    # replay the engine's streams for the first move
    rng = np.random.default_rng(3)
    rand = rng.integers(0, A, size=(G, S + 1), dtype=np.int32)
    eta = rng.dirichlet(np.full(A, 0.3), size=G).astype(np.float32)
    first_obs = torch.cat([Env(gi, 9, shape=(2, 8, 8), actions=A)._obs() for gi in range(G)])
    obs, policy32, value, action, reward, ended = eng.move()
    torch.cuda.synchronize()
    assert torch.equal(obs.cpu(), first_obs)
    hist = torch.cat([first_obs.repeat(1, K, 1, 1), torch.zeros(G, K, 8, 8)], dim=1)  # K copies + zero action planes
    with torch.no_grad():
        lat = torch.cat([h(hist[i:i + 1]) for i in range(G)]).reshape(G, -1)   # one game per call
    mcts = BatchedMCTS(eng.nets, G, S, 1, discount=0.998, noise_frac=0.25)
    res = mcts.run(lat.cuda(), torch.zeros(G, dtype=torch.int8, device="cuda"), torch.from_numpy(eta).cuda(),
                   torch.from_numpy(rand).cuda())
    np.testing.assert_allclose(policy32.cpu().numpy(), res.policy32.cpu().numpy(), rtol=1e-4, atol=1e-6)
    assert torch.equal(action.cpu(), res.policy32.argmax(dim=1).to(torch.int32).cpu())
    assert reward.cpu().tolist() == [float(a) for a in action.cpu().tolist()]
    # play on: episodes end where the simulators say and restart in place
    episodes = eng.generate(num_moves=9)
    assert len(episodes) >= G
    lengths = sorted({len(ep.chunks) for ep in episodes})
    assert set(lengths) <= {2, 3, 4, 5} and all(abs(float(c.policy.sum()) - 1) < 1e-5 for ep in episodes for c in ep.chunks)
    eng.env.close()


def _random_tower_nets(shape, A, g_blocks, f_blocks, act, seed):
    """g / f with residual blocks given as (out_channels, kernel_size) pairs, non-trivial BatchNorm affines."""
    from muzero_amd.config import DenseLayerConfig, DynamicsNetworkConfig, PredictionNetworkConfig, ResBlockConfig
    from muzero_amd.neural_networks import DynamicsNetwork, PredictionNetwork
    d = lambda n: DenseLayerConfig(out_features=n, activation="relu")  # noqa: E731
    rb = lambda c, k: ResBlockConfig(out_channels=c, kernel_size=k, stride=1, padding=(k - 1) // 2, activation=act,  # noqa: E731
                                     pool_kernel_size=1, pool_stride=1)
    torch.manual_seed(seed)
    g = DynamicsNetwork(shape, A, DynamicsNetworkConfig(res_net=[rb(c, k) for c, k in g_blocks], reward_net=[d(24)]))
    f = PredictionNetwork(shape, A, PredictionNetworkConfig(res_net=[rb(c, k) for c, k in f_blocks],
                                                            value_net=[d(16)], policy_net=[d(16)]))
    with torch.no_grad():
        for m in list(g.modules()) + list(f.modules()):
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
    return g, f


@pytest.mark.parametrize("shape,A,g_blocks,f_blocks,act", [
    ((4, 12, 12), 7, [(4, 3), (4, 5)], [(4, 1)], "relu"),            # 1008 positions per expansion call: several passes; identity skips; k = 1, 5
    ((40, 5, 5), 3, [(40, 3)], [(12, 3), (40, 3), (7, 3), (40, 3)], "tanh"),  # 40 channels: 10 output-channel passes; 4 blocks; tanh
    ((3, 7, 9), 4, [(9, 5), (3, 3)], [(3, 3)], "sigmoid"),            # odd H x W (no 16-byte rows), ends on C channels without res_final
    ((130, 6, 6), 2, [(3, 3)], [(3, 3)], "relu"),                     # the shipped form with a channel count that is no multiple of 4
    ((32, 6, 6), 10, [(3, 3)], [(3, 3)], "relu"),                     # 360 positions per expansion call: six 64-position passes, two rounds of wave sets
    ((8, 4, 4), 6, [(1, 5)], [(2, 3), (8, 1)], "relu"),               # matrix-product first block with k = 5 (26 columns), 96 positions per expansion call; a later k = 1 block; no res_final in f
])
def test_tower_kernel_general_shapes_match_torch(shape, A, g_blocks, f_blocks, act):
    """Towers the golden file does not hold (kernel sizes 1 and 5, other activations, many positions
    or channels per call, up to four blocks) against the torch modules evaluated call by call."""
    from muzero_amd.towers import TowerNetworks
    g, f = _random_tower_nets(shape, A, g_blocks, f_blocks, act, seed=sum(shape) + A)
    D = int(np.prod(shape))
    nets = TowerNetworks(g, f)
    B = 3
    torch.manual_seed(1)
    lat = torch.randn(B, *shape)
    act_t = torch.randint(0, A, (B,))
    with torch.no_grad():
        want_g1 = [g(lat[i:i + 1], act_t[i:i + 1]) for i in range(B)]
        want_gA = [g(lat[i:i + 1].repeat(A, 1, 1, 1), torch.arange(A)) for i in range(B)]
        want_f = [f(lat[i:i + 1]) for i in range(B)]
    x = lat.reshape(B, D).cuda().contiguous()
    nxt = torch.zeros(B, D, device="cuda")
    rew = torch.zeros(B, device="cuda")
    nets.dynamics(x, B, action=act_t.to(torch.int32).cuda(), next_latent=nxt, reward=rew)
    nxtA = torch.zeros(B * A, D, device="cuda")
    rewA = torch.zeros(B, A, device="cuda")
    nets.dynamics(x, B * A, rows=torch.arange(B, dtype=torch.int32, device="cuda").repeat_interleave(A), action=None,
                  next_latent=nxtA, reward=rewA)
    pol = torch.zeros(B, A, device="cuda")
    val = torch.zeros(B, device="cuda")
    nets.prediction(x, B, policy=pol, value=val)
    torch.cuda.synchronize()
    # 1e-5 relative, with the absolute floor at 1e-5 of the output's scale: BatchNorm divides by the
    # standard deviation of a few dozen values, which amplifies the last bit of near-zero outputs
    def close(got, want):
        want = want.numpy()
        np.testing.assert_allclose(got.cpu().numpy().reshape(want.shape), want, rtol=1e-5,
                                   atol=1e-5 * float(np.abs(want).max()))

    close(nxt, torch.cat([w[0] for w in want_g1]))
    close(rew, torch.cat([w[1] for w in want_g1])[:, 0])
    close(nxtA, torch.cat([w[0] for w in want_gA]))
    close(rewA, torch.stack([w[1][:, 0] for w in want_gA]))
    close(pol, torch.cat([w[0] for w in want_f]))
    close(val, torch.cat([w[1] for w in want_f])[:, 0])


def test_more_groups_than_compute_units():
    """2048 reference calls in one launch -- batch size 1 (g step, f) and batch size A (expansion):
    the build of the tower kernel that shares a compute unit between two workgroups, each workgroup
    taking four calls with their own batch statistics -- against the torch modules call by call."""
    from muzero_amd.towers import TowerNetworks
    shape, A = (16, 4, 4), 3
    g, f = _random_tower_nets(shape, A, [(3, 3)], [(3, 3)], "relu", seed=11)
    D = int(np.prod(shape))
    nets = TowerNetworks(g, f)
    B = 2048
    torch.manual_seed(4)
    lat = torch.randn(B, *shape)
    act_t = torch.randint(0, A, (B,))
    with torch.no_grad():
        want_g = [g(lat[i:i + 1], act_t[i:i + 1]) for i in range(B)]
        want_gA = [g(lat[i:i + 1].repeat(A, 1, 1, 1), torch.arange(A)) for i in range(B)]
        want_f = [f(lat[i:i + 1]) for i in range(B)]
    x = lat.reshape(B, D).cuda().contiguous()
    nxt = torch.zeros(B, D, device="cuda")
    rew = torch.zeros(B, device="cuda")
    nets.dynamics(x, B, action=act_t.to(torch.int32).cuda(), next_latent=nxt, reward=rew)
    nxtA = torch.zeros(B * A, D, device="cuda")
    rewA = torch.zeros(B, A, device="cuda")
    nets.dynamics(x, B * A, rows=torch.arange(B, dtype=torch.int32, device="cuda").repeat_interleave(A), action=None,
                  next_latent=nxtA, reward=rewA)
    pol = torch.zeros(B, A, device="cuda")
    val = torch.zeros(B, device="cuda")
    nets.prediction(x, B, policy=pol, value=val)
    torch.cuda.synchronize()

    def close(got, want):
        want = want.numpy()
        np.testing.assert_allclose(got.cpu().numpy().reshape(want.shape), want, rtol=1e-5,
                                   atol=1e-5 * float(np.abs(want).max()))

    close(nxt, torch.cat([w[0] for w in want_g]))
    close(rew, torch.cat([w[1] for w in want_g])[:, 0])
    close(nxtA, torch.cat([w[0] for w in want_gA]))
    close(rewA, torch.stack([w[1][:, 0] for w in want_gA]))
    close(pol, torch.cat([w[0] for w in want_f]))
    close(val, torch.cat([w[1] for w in want_f])[:, 0])


def test_shipped_shape_matches_torch():
    """The shipped g / f (latent [256, 6, 6], fc_merge 18432 -> 9216, one 256 -> 3 block, res_final, heads on
    9216 inputs) at full size against the torch modules on the CPU: 1e-5 of each output's scale."""
    from muzero_amd.towers import TowerNetworks
    from tools.bench_towers import build_nets
    shape, A = (256, 6, 6), 5
    g, f = build_nets(shape, A)
    with torch.no_grad():
        for m in list(g.modules()) + list(f.modules()):
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
    D = 9216
    nets = TowerNetworks(g, f)
    B = 2
    torch.manual_seed(2)
    lat = torch.randn(B, *shape)
    with torch.no_grad():
        want_gA = [g(lat[i:i + 1].repeat(A, 1, 1, 1), torch.arange(A)) for i in range(B)]
        want_g1 = [g(lat[i:i + 1], torch.tensor([i + 1])) for i in range(B)]
        want_f = [f(lat[i:i + 1]) for i in range(B)]
    x = lat.reshape(B, D).cuda().contiguous()
    nxtA = torch.zeros(B * A, D, device="cuda")
    rewA = torch.zeros(B, A, device="cuda")
    nets.dynamics(x, B * A, rows=torch.arange(B, dtype=torch.int32, device="cuda").repeat_interleave(A), action=None,
                  next_latent=nxtA, reward=rewA)
    nxt = torch.zeros(B, D, device="cuda")
    rew = torch.zeros(B, device="cuda")
    nets.dynamics(x, B, action=torch.tensor([1, 2], dtype=torch.int32, device="cuda"), next_latent=nxt, reward=rew)
    pol = torch.zeros(B, A, device="cuda")
    val = torch.zeros(B, device="cuda")
    nets.prediction(x, B, policy=pol, value=val)
    torch.cuda.synchronize()

    def close(got, want):
        want = want.numpy()
        np.testing.assert_allclose(got.cpu().numpy().reshape(want.shape), want, rtol=1e-5,
                                   atol=1e-5 * float(np.abs(want).max()))

    close(nxtA, torch.cat([w[0] for w in want_gA]))
    close(rewA, torch.stack([w[1][:, 0] for w in want_gA]))
    close(nxt, torch.cat([w[0] for w in want_g1]))
    close(rew, torch.cat([w[1] for w in want_g1])[:, 0])
    close(pol, torch.cat([w[0] for w in want_f]))
    close(val, torch.cat([w[1] for w in want_f])[:, 0])


def test_graph_replay_equals_launch_by_launch():
    """The recorded search (one HIP graph of all simulation steps, two streams) leaves the same
    pool, bit for bit, as launching its kernels one by one: the first search (run launch by launch),
    the next one (recorded, then replayed), and the same pair after a weight refresh."""
    from muzero_amd.engine import BatchedMCTS
    from muzero_amd.towers import TowerNetworks
    g, f, A, shape = load_tower_nets(0)
    D = int(np.prod(shape))
    nets = TowerNetworks(g, f)
    G, S = 9, 14
    rng = np.random.default_rng(5)

    def inputs():
        lat = torch.from_numpy(rng.standard_normal((G, D)).astype(np.float32)).cuda()
        eta = torch.from_numpy(rng.dirichlet(np.full(A, 0.3), size=G).astype(np.float32)).cuda()
        rand = torch.from_numpy(rng.integers(0, A, size=(G, S + 1)).astype(np.int32)).cuda()
        return lat, torch.ones(G, dtype=torch.int8, device="cuda"), eta, rand

    eager = BatchedMCTS(nets, G, S, 2, discount=0.998, noise_frac=0.25)
    eager.use_graph = False
    graphed = BatchedMCTS(nets, G, S, 2, discount=0.998, noise_frac=0.25)
    lean = BatchedMCTS(nets, G, S, 2, discount=0.998, noise_frac=0.25, keep_latents=False)  # children kept as bottleneck values only
    assert not lean.keep_latents
    for round_ in range(4):
        if round_ == 2:
            with torch.no_grad():
                g.fc_merge.bias.add_(0.01)
            nets.refresh()
        args = inputs()
        a, b, c = eager.run(*args), graphed.run(*args), lean.run(*args)
        torch.cuda.synchronize()
        for name in ("visit", "value_sum", "prior", "reward", "first_child"):
            assert torch.equal(getattr(eager.pool, name), getattr(lean.pool, name)), (round_, name, "lean")
        assert torch.equal(a.policy, c.policy) and torch.equal(eager.t_pool, lean.t_pool)
        assert float(lean.pool.latent[:, 1:].abs().max()) == 0.0
        assert graphed.replays == (0, 1, 1, 2)[round_] and (graphed._graph is not None) == (round_ > 0)
        for name in ("visit", "value_sum", "prior", "reward", "first_child", "latent"):
            assert torch.equal(getattr(eager.pool, name), getattr(graphed.pool, name)), (round_, name)
        assert torch.equal(a.policy, b.policy) and torch.equal(a.value, b.value) and torch.equal(a.visits, b.visits)


@pytest.mark.parametrize("B", [5, 256, 1000])
def test_policy_head_with_post_processing_in_one_launch(B):
    """mz_mlp_forward_post = mz_mlp_forward followed by mz_policy_post, bit for bit: extra softmax
    passes 0..2, the first-maximum argmax, output only / argmax only."""
    import ctypes as C
    from muzero_amd import _hip
    from muzero_amd._hip import ptr
    from muzero_amd.towers import TowerNetworks
    g, f, A, shape = load_tower_nets(0)
    nets = TowerNetworks(g, f)
    assert nets.bottleneck_f
    dt = nets.bottleneck_f
    lib = _hip.lib()
    st = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator(device="cuda").manual_seed(B)
    t = torch.randn(B, dt, device="cuda", generator=gen)
    t[1] = t[0]  # equal rows: equal outputs
    raw = torch.empty(B, A, device="cuda")
    _hip.check(lib.mz_mlp_forward(C.byref(nets.policy_t), ptr(nets.blob), ptr(t), dt, None, ptr(raw), A, B, st), "fwd")
    for times in (0, 1, 2):
        want = torch.empty(B, A, device="cuda")
        want_arg = torch.empty(B, dtype=torch.int32, device="cuda")
        _hip.check(lib.mz_policy_post(ptr(raw), B, A, times, ptr(want), ptr(want_arg), st), "post")
        got = torch.full((B, A), -1.0, device="cuda")
        got_arg = torch.full((B,), -1, dtype=torch.int32, device="cuda")
        _hip.check(lib.mz_mlp_forward_post(C.byref(nets.policy_t), ptr(nets.blob), ptr(t), dt, None, ptr(got), A, B,
                                           times, ptr(got_arg), st), "fwd_post")
        torch.cuda.synchronize()
        assert torch.equal(got, want) and torch.equal(got_arg, want_arg)
        only_arg = torch.full((B,), -1, dtype=torch.int32, device="cuda")
        _hip.check(lib.mz_mlp_forward_post(C.byref(nets.policy_t), ptr(nets.blob), ptr(t), dt, None, None, A, B, times,
                                           ptr(only_arg), st), "argmax only")
        only_out = torch.full((B, A), -1.0, device="cuda")
        _hip.check(lib.mz_mlp_forward_post(C.byref(nets.policy_t), ptr(nets.blob), ptr(t), dt, None, ptr(only_out), A, B,
                                           times, None, st), "output only")
        torch.cuda.synchronize()
        assert torch.equal(only_arg, want_arg) and torch.equal(only_out, want)
    assert lib.mz_mlp_forward_post(C.byref(nets.policy_t), ptr(nets.blob), ptr(t), dt, None, None, A, B, 0, None, st) != 0


def test_tower_search_at_benchmarked_size():
    """The configuration the tower numbers are quoted on -- latent [256, 6, 6], 5 actions, 50
    simulations, 256 trees, bottleneck-form lock-step search -- with and without stored latents,
    launch by launch and replayed from the recorded HIP graph:
      (i)  the search's own network outputs replayed through the CPU tree oracle give every tree bit
           for bit, and the graph replay leaves the same pool as the traced launch-by-launch run;
      (ii) the prior / reward / rollout streams of 8 trees equal the torch modules called one
           reference call at a time (src/search/expansion.py:25-33, src/search/simulation.py:31-44)
           on the search's own latents, to 1e-5 of each output's scale."""
    from muzero_amd.engine import BatchedMCTS
    from muzero_amd.towers import TowerNetworks
    from tools.bench_towers import build_nets
    shape, A, G, S, depth = (256, 6, 6), 5, 256, 50, 1
    D = 9216
    g, f = build_nets(shape, A)
    with torch.no_grad():
        for m in list(g.modules()) + list(f.modules()):
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
    nets = TowerNetworks(g, f, fuse_merge_conv=False)
    fused = TowerNetworks(g, f)   # fc_merge + the first convolution of g's tower in one kernel (searches without stored latents)
    assert fused.fused_g and not nets.fused_g
    def searched(networks, G, keep_flag, seed):
        rng = np.random.default_rng(seed)
        lat = rng.standard_normal((G, D)).astype(np.float32)
        eta = rng.dirichlet(np.full(A, 0.3), size=G).astype(np.float32)
        rand = rng.integers(0, A, size=(G, S + 1)).astype(np.int32)
        to_play = np.zeros(G, np.int8)   # single-player simulators (CarRacing): to_play 0
        dev = [torch.from_numpy(x).cuda() for x in (lat, to_play, eta, rand)]
        mcts = BatchedMCTS(networks, G, S, depth, discount=0.998, noise_frac=0.25, keep_latents=keep_flag)
        assert mcts.t_pool is not None and mcts.keep_latents == keep_flag, "the bottleneck form is not in use"
        trace = {}
        res = mcts.run(*dev, trace=trace)          # launch by launch (tracing)
        torch.cuda.synchronize()
        pool = mcts.pool
        prior_s = np.stack([p.cpu().numpy() for p in trace["prior"]], axis=1)
        rollout = np.stack([r.cpu().numpy() for r in trace["rollout"]], axis=1)
        leaf = np.stack([x.cpu().numpy() for x in trace["leaf"]], axis=1)
        child = np.stack([x.cpu().numpy() for x in trace["child"]], axis=1)
        reward_nodes = pool.reward.cpu().numpy()
        root_prior = torch.empty(G, A, device="cuda")
        mcts.nets.prediction(dev[0], G, softmax_times=1, policy=root_prior)
        prior_stream = np.concatenate([root_prior.cpu().numpy()[:, None], prior_s], axis=1)
        reward_stream = np.stack([reward_nodes[:, 1 + e * A:1 + (e + 1) * A] for e in range(S + 1)], axis=1)
        ref = otree.run_batch(A, S, depth, to_play, prior_stream, reward_stream, rand, rollout, eta, 0.25, 0.998)
        assert np.array_equal(leaf, ref["leaf"]) and np.array_equal(child, ref["child"])
        for name in ("visit", "value_sum", "prior", "first_child", "parent"):
            assert np.array_equal(getattr(pool, name).cpu().numpy(), ref[name]), name
        assert np.array_equal(res.value.cpu().numpy(), ref["utility"])
        np.testing.assert_allclose(res.policy.cpu().numpy(), ref["policy"], rtol=1e-12)
        stats = {n: getattr(pool, n).clone() for n in ("visit", "value_sum", "prior", "reward", "parent", "first_child")}
        # the same search twice more without tracing: the first records the graph, both replay it
        for _ in range(2):
            res2 = mcts.run(*dev)
            torch.cuda.synchronize()
            for n, want in stats.items():
                assert torch.equal(getattr(pool, n), want), f"graph replay: {n}"
            assert torch.equal(res2.policy32, res.policy32) and torch.equal(res2.value, res.value)
        assert mcts.replays == 2, "the recorded graph was not used"
        return dict(prior=prior_stream, reward=reward_stream, rollout=rollout, leaf=leaf, child=child,
                    latent=pool.latent if keep_flag else None, t_pool=mcts.t_pool.clone(), mcts=mcts)

    pools = {keep: searched(nets, G, keep, 17) for keep in (True, False)}
    keep_mcts = pools[True]["mcts"]   # keeps pool.latent alive for (ii)
    pools[False].pop("mcts")
    # storing the latents or not changes nothing the search computes
    for n in ("prior", "reward", "rollout", "leaf", "child"):
        assert np.array_equal(pools[True][n], pools[False][n]), n
    assert torch.equal(pools[True]["t_pool"], pools[False]["t_pool"])

    # The kernels used from a thousand reference calls per launch on (one wave per call for the small blocks;
    # fc_merge + the first convolution of g in one kernel) compute the same numbers in another order: at that
    # size, once with the separate kernels and once fused, each run checked as above; wherever a tree's descents
    # agree its streams do to 1e-5 of their scale, and a tree's first disagreeing descent is a near-tie.
    G2 = 1024
    assert fused.fuse_min_rows <= G2
    plain_big = searched(nets, G2, False, 18)
    fused_big = searched(fused, G2, False, 18)
    same = np.array([np.array_equal(fused_big["leaf"][gi], plain_big["leaf"][gi]) and
                     np.array_equal(fused_big["child"][gi], plain_big["child"][gi]) for gi in range(G2)])
    assert same.mean() >= 0.9, f"only {int(same.sum())} of {G2} trees took the same descents"
    for n in ("prior", "reward", "rollout"):
        a, b = fused_big[n][same], plain_big[n][same]
        assert not np.array_equal(a, b), "the fused kernels were not used"
        np.testing.assert_allclose(a, b, rtol=1e-5, atol=max(1e-5 * float(np.abs(b).max()), TOL["atol"]), err_msg=n)
    for gi in np.flatnonzero(~same):
        first = int(np.flatnonzero((fused_big["leaf"][gi] != plain_big["leaf"][gi]) |
                                   (fused_big["child"][gi] != plain_big["child"][gi]))[0])
        for n in ("prior", "reward"):   # everything computed before the first differing descent agrees
            a, b = fused_big[n][gi, :first + 1], plain_big[n][gi, :first + 1]
            np.testing.assert_allclose(a, b, rtol=1e-5, atol=max(1e-5 * float(np.abs(b).max()), TOL["atol"]), err_msg=f"{n} tree {gi}")
    del plain_big, fused_big

    # ---- (ii) the torch modules, one reference call at a time, on the GPU (rocBLAS / MIOpen fp32)
    gt, ft = g.cuda(), f.cuda()
    P = pools[True]
    latent = keep_mcts.pool.latent   # [G, cap, D]
    acts = torch.arange(A, device="cuda")

    def close(got, want, what):
        want = want.detach().cpu().numpy()
        # 1e-5 relative; absolute floor: 1e-5 of the output's scale, and never below the 2e-6 the
        # reference-vector comparisons of this file use for head outputs of a few hundredths
        np.testing.assert_allclose(np.asarray(got).reshape(want.shape), want, rtol=1e-5,
                                   atol=max(1e-5 * float(np.abs(want).max()), TOL["atol"]), err_msg=what)

    near_ties = 0
    with torch.no_grad():
        for gi in range(0, G, G // 8):
            for s in list(range(0, S, 7)) + [S - 1]:
                lf, ch = int(P["leaf"][gi, s]), int(P["child"][gi, s])
                x = latent[gi, lf].reshape(1, *shape)
                close(P["prior"][gi, s + 1], torch.softmax(ft(x)[0], dim=1)[0], f"prior tree {gi} sim {s}")
                nxt, rew = gt(x.repeat(A, 1, 1, 1), acts)   # ONE call: batch statistics over the A children
                base = 1 + (s + 1) * A
                close(P["reward"][gi, s + 1], rew[:, 0], f"reward tree {gi} sim {s}")
                close(latent[gi, base:base + A].cpu().numpy(), nxt.reshape(A, D), f"child latents tree {gi} sim {s}")
                cur = latent[gi, ch].reshape(1, *shape)
                pol = ft(cur)[0][0]
                top2 = pol.topk(2).values
                if float(top2[0] - top2[1]) <= 1e-5 * float(top2[0]):
                    near_ties += 1     # the greedy action is decided by rounding: nothing to compare
                    continue
                cur, r = gt(cur, pol.argmax().reshape(1))
                v = ft(cur)[1]
                close(P["rollout"][gi, s], torch.stack([r[0, 0], v[0, 0]]), f"rollout tree {gi} sim {s}")
    assert near_ties <= 2
    g.cpu(), f.cpu()


def test_one_wave_per_call_kernel_matches_the_general_kernel():
    """Small blocks (f's tower on the bottleneck values: 3 -> 3 channels on 6 x 6) run on a one-wave-per-call kernel;
    mz_tower_thin_min_calls moves the call count from which it is used, so the general tower kernel can take the
    same rows (one launch of 2048 calls against four launches of 512 with the threshold above them): outputs and
    exported batch statistics agree to 1e-6 of scale (the two kernels add the window in different orders), for
    batch-size-1 calls and for calls of 5 rows; a handful of calls (3, and 1) go through the one-wave kernel too."""
    import ctypes as C
    from muzero_amd import _hip
    from muzero_amd._hip import ptr
    from muzero_amd.towers import TowerNetworks
    from tools.bench_towers import build_nets
    g, f = build_nets((256, 6, 6), 5)
    with torch.no_grad():
        for m in f.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
    nets = TowerNetworks(g, f)
    tower = nets.tower_f_t
    assert tower is not None and tower.n_blocks == 1
    lib = _hip.lib()
    st = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator(device="cuda").manual_seed(5)
    for group, calls in ((1, 2048), (5, 1024)):
        B = calls * group
        src = torch.randn(B + 7, 108, device="cuda", generator=gen)
        rows = torch.randperm(B + 7, device="cuda", generator=gen)[:B].to(torch.int32)
        if group > 1:   # the rows of a call share one source latent in the expansion; here they simply differ
            rows = rows.contiguous()
        out_rows = torch.randperm(B, device="cuda", generator=gen).to(torch.int32)
        n_stats = lib.mz_tower_stats_floats(C.byref(tower), calls)

        def run(chunk_calls, n_calls=calls):
            out = torch.zeros(B, 108, device="cuda")
            stats = []
            for c0 in range(0, n_calls, chunk_calls):
                r0, nb = c0 * group, chunk_calls * group
                st_c = torch.zeros(lib.mz_tower_stats_floats(C.byref(tower), chunk_calls), device="cuda")
                _hip.check(lib.mz_tower_forward_stats(C.byref(tower), ptr(src), 108, ptr(rows[r0:r0 + nb].contiguous()), None, 0, None,
                                                      None, 0, ptr(out_rows[r0:r0 + nb].contiguous()), ptr(out), 108, nb, group,
                                                      ptr(st_c), st), "mz_tower_forward_stats")
                stats.append(st_c.view(-1, 2, chunk_calls, 3))   # [layer][mean | variance][call][channel]
            torch.cuda.synchronize()
            return out, torch.cat(stats, dim=2)

        old = lib.mz_tower_thin_min_calls(1)
        try:
            thin, thin_stats = run(calls)
            few, few_stats = run(3, 6)     # two launches of three calls
            one, _ = run(1, 1)
            lib.mz_tower_thin_min_calls(1 << 30)
            general, general_stats = run(512 // (2 if group > 1 else 1))
        finally:
            lib.mz_tower_thin_min_calls(old)
        assert thin_stats.numel() == n_stats
        scale = float(general.abs().max())
        assert float((thin - general).abs().max()) <= 1e-6 * scale, (group, float((thin - general).abs().max()), scale)
        np.testing.assert_allclose(thin_stats.cpu().numpy(), general_stats.cpu().numpy(), rtol=1e-5, atol=1e-7)
        assert not torch.equal(thin, torch.zeros_like(thin))
        touched = out_rows[:6 * group].long()
        assert torch.equal(few[touched], thin[touched]) and torch.equal(one[touched[:group]], thin[touched[:group]])
        assert torch.equal(few_stats, thin_stats[:, :, :6])


def test_running_statistics_through_the_fused_form():
    """track_running_stats with fc_merge + the first convolution of g in one kernel: after an expansion launch
    (1024 calls of 5 rows) and a rollout-step launch (1024 calls of 1 row) the BatchNorm buffers written back into
    the torch modules equal those the separate kernels leave, to 1e-5 (both export every call's statistics and run
    the same momentum chain in call order)."""
    import copy
    from muzero_amd.towers import TowerNetworks
    from tools.bench_towers import build_nets
    A, calls = 5, 1024
    g1, f1 = build_nets((256, 6, 6), A)
    with torch.no_grad():
        for m in g1.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
    g2, f2 = copy.deepcopy(g1), copy.deepcopy(f1)
    plain = TowerNetworks(g1, f1, track_running_stats=True, fuse_merge_conv=False)
    fused = TowerNetworks(g2, f2, track_running_stats=True)
    assert fused.fused_g and fused.fuse_min_rows <= calls
    gen = torch.Generator(device="cuda").manual_seed(11)
    t_src = torch.randn(calls + 3, 108, device="cuda", generator=gen)
    rows = torch.randperm(calls + 3, device="cuda", generator=gen)[:calls].to(torch.int32)
    act = torch.randint(0, A, (calls,), device="cuda", generator=gen).to(torch.int32)
    outs = []
    for nets in (plain, fused):
        t_out = torch.zeros(calls * A, 108, device="cuda")
        reward = torch.zeros(calls * A, device="cuda")
        rows_rep = rows.repeat_interleave(A).contiguous()
        nets.dynamics(None, calls * A, rows=rows_rep, action=None, t_in=(t_src, rows_rep), t_out=t_out, reward=reward,
                      want_latent=False)
        t_step = torch.zeros(calls, 108, device="cuda")
        r_step = torch.zeros(calls, device="cuda")
        nets.dynamics(None, calls, rows=rows, action=act, t_in=(t_src, rows), t_out=t_step, reward=r_step, want_latent=False)
        nets.sync_running_stats()
        torch.cuda.synchronize()
        outs.append((t_out, reward, t_step, r_step))
    for a, b, what in zip(outs[0], outs[1], ("expansion t", "expansion reward", "step t", "step reward")):
        scale = float(a.abs().max())
        assert float((a - b).abs().max()) <= 1e-5 * scale, what
        assert not torch.equal(a, b) or "reward" in what, "the fused kernels were not used"
    checked = 0
    for (n1, m1), (_, m2) in zip(g1.named_modules(), g2.named_modules()):
        if isinstance(m1, torch.nn.BatchNorm2d):
            assert int(m1.num_batches_tracked) == int(m2.num_batches_tracked) == 2 * calls, n1
            np.testing.assert_allclose(m2.running_mean.cpu().numpy(), m1.running_mean.cpu().numpy(), rtol=1e-5, atol=1e-6, err_msg=n1)
            np.testing.assert_allclose(m2.running_var.cpu().numpy(), m1.running_var.cpu().numpy(), rtol=1e-5, atol=1e-7, err_msg=n1)
            checked += 1
    assert checked >= 3


@pytest.mark.parametrize("ni,ti", TREES[:2])
def test_tower_running_statistics_follow_the_reference_calls(ni, ti):
    """track_running_stats: after a search the BatchNorm buffers of g and f equal those the
    reference's train-mode calls leave in the torch modules (network_builder.py:94,104,118) -- the
    per-node CPU port makes the same calls at the same batch sizes (3 x f at batch 1, g at batch A, g
    at batch 1 per simulation), torch updates its buffers as a side effect; here the tower kernels
    export each call's statistics and the momentum chain runs on the device.  Same recorded search
    (root, eta, child draws) on both sides, one tree; then again through the HIP graph replay."""
    import copy
    from oracle import pyref
    from muzero_amd.engine import BatchedMCTS
    from muzero_amd.towers import TowerNetworks
    k = f"net{ni}_t{ti}_"
    g, f, A, shape = load_tower_nets(ni)
    D = int(np.prod(shape))
    S, depth = int(Z[k + "S"]), int(Z[k + "depth"])
    noise, disc, tp = float(Z[k + "noise_frac"]), float(Z[k + "discount"]), int(Z[k + "to_play"])
    lat = Z[k + "root_latent"].reshape(1, D).astype(np.float32)
    eta, rand = Z[k + "eta"].astype(np.float32), Z[k + "rand"].astype(np.int32)
    g_ref, f_ref = copy.deepcopy(g).train(), copy.deepcopy(f).train()
    nets = TowerNetworks(g, f, track_running_stats=True)
    mcts = BatchedMCTS(nets, 1, S, depth, discount=disc, noise_frac=noise)
    searches = 3  # the second warms the launch graph up, the third replays it
    for _ in range(searches):
        mcts.run(torch.from_numpy(lat).cuda(), torch.tensor([tp], dtype=torch.int8).cuda(), torch.from_numpy(eta[None]).cuda(),
                 torch.from_numpy(rand[None]).cuda())
        draws = iter(rand.tolist())
        with torch.no_grad():
            ev = pyref.TorchEvaluator(g_ref, f_ref, torch.arange(A).long(), depth)
            _, _, nodes = pyref.run_mcts(torch.from_numpy(lat).reshape(1, *shape), tp, ev, A, S, discount=disc, noise_frac=noise,
                                         eta=eta, rand_index=lambda n: next(draws))
    torch.cuda.synchronize()
    assert mcts.replays >= 1
    assert np.array_equal(mcts.pool.visit.cpu().numpy()[0, :len(nodes)], [nd.visit for nd in nodes])  # the same tree
    nets.sync_running_stats()
    checked = 0
    for mine, ref in ((g, g_ref), (f, f_ref)):
        for (name, a), (_, b) in zip(mine.named_buffers(), ref.named_buffers()):
            if name.endswith("num_batches_tracked"):
                assert int(a) == int(b) > 0, name
            else:
                np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-4, atol=1e-6, err_msg=name)
                checked += 1
    assert checked >= 8
    # the parameters were not touched
    assert all(torch.equal(p, q) for p, q in zip(list(g.parameters()) + list(f.parameters()),
                                                 list(g_ref.parameters()) + list(f_ref.parameters())))


@pytest.mark.parametrize("depth", [1, 2])
def test_one_chain_recorded_search_equals_launch_by_launch(depth):
    """Shipped network shapes (latent [256, 6, 6]), nodes kept as bottleneck values: the recorded search runs a step as
    one chain of launches -- one-launch tree step (mz_tree_commit_backup_select), the children's rewards together with
    the first rollout step's, reward and value heads side by side -- whereas the launch-by-launch search uses the
    separate kernels.  Same pools, bottleneck values and results, bit for bit, with a look-ahead of 1 and of 2 and a
    tree count that fills no tile."""
    from muzero_amd.engine import BatchedMCTS
    from muzero_amd.towers import TowerNetworks
    from tools.bench_towers import build_nets
    A = 5
    g, f = build_nets((256, 6, 6), A)
    nets = TowerNetworks(g, f)
    assert nets.fused_g
    G, S, D = 40, 8, 256 * 36
    rng = np.random.default_rng(21 + depth)
    eager = BatchedMCTS(nets, G, S, depth, discount=0.997, noise_frac=0.25, keep_latents=False)
    eager.use_graph = False
    graphed = BatchedMCTS(nets, G, S, depth, discount=0.997, noise_frac=0.25, keep_latents=False)
    assert graphed._forked() and graphed.step_schedule == "chain" and graphed.fuse_tree_kernels
    for round_ in range(3):
        lat = torch.from_numpy(rng.standard_normal((G, D)).astype(np.float32)).cuda()
        eta = torch.from_numpy(rng.dirichlet(np.full(A, 0.3), size=G).astype(np.float32)).cuda()
        rand = torch.from_numpy(rng.integers(0, A, size=(G, S + 1)).astype(np.int32)).cuda()
        to_play = torch.ones(G, dtype=torch.int8, device="cuda")
        a, b = eager.run(lat, to_play, eta, rand), graphed.run(lat, to_play, eta, rand)
        torch.cuda.synchronize()
        assert graphed.replays == (0, 1, 2)[round_]
        for name in ("visit", "value_sum", "prior", "reward", "first_child"):
            assert torch.equal(getattr(eager.pool, name), getattr(graphed.pool, name)), (round_, name)
        cap = eager.pool.cap
        assert torch.equal(eager.t_pool[:G * cap], graphed.t_pool[:G * cap])
        assert torch.equal(a.policy, b.policy) and torch.equal(a.value, b.value) and torch.equal(a.visits, b.visits)
        assert int(a.visits.sum()) == G * S


def test_root_expansion_from_the_representation_networks_bottleneck_values():
    """A root latent is h's final 1x1 convolution of a few pooled channels (neural_network.py:51,60-61), the same kind
    of affine image of 108 values as g's outputs: with `root_final` the root expansion runs in the bottleneck form
    from those values (weights composed with that convolution) instead of on the 9216-wide latent.  Priors, rewards
    and the children's bottleneck values of the root expansion agree with the plain form to 1e-5 of scale; whole
    searches build the same trees wherever no near-tie decides, and run without the root latents at all."""
    from muzero_amd.engine import BatchedMCTS
    from muzero_amd.towers import TowerNetworks
    from tools.bench_towers import build_nets
    A, G, S = 5, 48, 10
    g, f = build_nets((256, 6, 6), A)
    torch.manual_seed(5)
    fin = torch.nn.Conv2d(3, 256, 1).cuda()
    rng = np.random.default_rng(3)
    root_t = torch.from_numpy(rng.standard_normal((G, 3, 6, 6)).astype(np.float32)).cuda()
    with torch.no_grad():
        lat = fin(root_t).reshape(G, -1).contiguous()
    eta = torch.from_numpy(rng.dirichlet(np.full(A, 0.3), size=G).astype(np.float32)).cuda()
    rand = torch.from_numpy(rng.integers(0, A, size=(G, S + 1)).astype(np.int32)).cuda()
    to_play = torch.ones(G, dtype=torch.int8, device="cuda")
    nets = TowerNetworks(g, f, root_final=fin)
    assert nets.has_root_form and not TowerNetworks(g, f).has_root_form
    plain = BatchedMCTS(nets, G, S, 1, discount=0.997, noise_frac=0.25, keep_latents=False)
    rooted = BatchedMCTS(nets, G, S, 1, discount=0.997, noise_frac=0.25, keep_latents=False)
    assert rooted.root_form_ok(root_t) and not plain.root_form_ok(None)
    plain.begin(lat, to_play, eta, rand[:, 0])
    rooted.begin(None, to_play, eta, rand[:, 0], root_t=root_t)
    torch.cuda.synchronize()
    cap = plain.pool.cap
    kids = (torch.arange(G, device="cuda")[:, None] * cap + 1 + torch.arange(A, device="cuda")[None, :]).reshape(-1)
    for name, a, b in (("prior", rooted.pool.prior[:, 1:1 + A], plain.pool.prior[:, 1:1 + A]),
                       ("reward", rooted.pool.reward[:, 1:1 + A], plain.pool.reward[:, 1:1 + A]),
                       ("children", rooted.t_pool[kids], plain.t_pool[kids])):
        a, b = a.double().cpu().numpy(), b.double().cpu().numpy()
        np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-5 * float(np.abs(b).max()), err_msg=name)
    assert float(rooted.pool.latent.abs().max()) == 0.0   # no latent was ever stored
    ra = plain.run(lat, to_play, eta, rand)
    rb = rooted.run(None, to_play, eta, rand, root_t=root_t)
    rb2 = rooted.run(None, to_play, eta, rand, root_t=root_t)   # recorded graph
    torch.cuda.synchronize()
    assert int(rb.visits.sum()) == G * S and torch.equal(rb.visits, rb2.visits) and torch.equal(rb.policy, rb2.policy)
    same = (ra.visits == rb.visits).all(dim=1)
    assert float(same.float().mean()) >= 0.85, f"only {int(same.sum())} of {G} trees agree"
    np.testing.assert_allclose(rb.value[same].cpu().numpy(), ra.value[same].cpu().numpy(), rtol=1e-4, atol=1e-6)


def test_selfplay_starts_tower_searches_from_the_pooled_input_of_hs_final_convolution():
    """Self-play with a representation network that ends in a 1x1 convolution of 3 channels and the shipped tower
    shape: the engine hands the search the pooled 3-channel tensor (checked against the module's own input of that
    convolution, pooled by torch) and the moves agree with an engine that starts from the 256-channel latents."""
    from muzero_amd.config import (ConvLayerConfig, MCTSConfig, RepresentationNetworkConfig, TrainingDataGeneratorConfig)
    from muzero_amd.environments.host_pool import HostEnvPool
    from muzero_amd.neural_networks import RepresentationNetwork
    from muzero_amd.selfplay import SelfPlayEngine
    from tests.test_host_pool import CountingEnv
    from tools.bench_towers import build_nets
    A, G, S, K = 5, 24, 10, 2
    shape = (256, 6, 6)
    g, f = build_nets(shape, A)
    torch.manual_seed(11)
    h = RepresentationNetwork((2, 24, 24), shape, RepresentationNetworkConfig(history_length=K, downsample=[
        ConvLayerConfig(out_channels=3, kernel_size=2, stride=2, padding=0, activation="relu")], res_net=[]))
    assert h.final_conv.in_channels == 3 and h.final_conv.kernel_size == (1, 1)

    class Env(CountingEnv):
        def get_to_play(self):
            return 0

        def _obs(self):
            gen = torch.Generator().manual_seed(self.gid * 1000 + self.t)
            return torch.rand((1,) + self.shape, generator=gen)

    cfg = TrainingDataGeneratorConfig(
        num_episodes=1, max_steps_per_episode=50, total_time=0, epsilon=0.0, epsilon_decay=1.0, min_epsilon=0.0,
        mcts=MCTSConfig(selection_strategy="puct", max_iterations=S, max_time=0, model_look_ahead=1,
                        discount_factor=0.998, dirichlet_alpha=0.3, noise_frac=0.25))
    mk = lambda: SelfPlayEngine(h, g, f, G, cfg, seed=3,  # noqa: E731
                                env=HostEnvPool([Env(gi, 30, shape=(2, 24, 24), actions=A) for gi in range(G)], workers=2))
    rooted, plain = mk(), mk()
    assert rooted.nets.has_root_form
    plain.nets.has_root_form = False
    seen = []
    hook = h.final_conv.register_forward_pre_hook(lambda m, x: seen.append(x[0].detach()))
    first_obs = torch.cat([Env(gi, 30, shape=(2, 24, 24), actions=A)._obs() for gi in range(G)])
    hist = torch.cat([first_obs.repeat(1, K, 1, 1), torch.zeros(G, K, 24, 24)], dim=1)
    with torch.no_grad():
        h(hist)
    hook.remove()
    want_t = torch.nn.functional.adaptive_avg_pool2d(seen[0], (6, 6))
    pa = plain.move()
    ra = rooted.move()
    torch.cuda.synchronize()
    assert plain.h.root_t is None and rooted.h.root_t is not None
    np.testing.assert_allclose(rooted.h.root_t.cpu().numpy(), want_t.numpy(), rtol=1e-5, atol=1e-6)
    assert float(rooted.mcts.pool.latent.abs().max()) == 0.0   # the 256-channel latents never reached the search
    # same trees wherever no near-tie decides: policies within rounding, then equal values and actions
    same = ((pa[1] - ra[1]).abs() <= 1e-5).all(dim=1)
    assert float(same.float().mean()) >= 0.85, f"only {int(same.sum())} of {G} games agree"
    np.testing.assert_allclose(ra[2][same].cpu().numpy(), pa[2][same].cpu().numpy(), rtol=1e-4, atol=1e-6)
    assert torch.equal(ra[3][same], pa[3][same])
    # a weight update of h reaches the composed root weights without an explicit refresh of the towers
    with torch.no_grad():
        h.final_conv.weight.mul_(1.5)
    v0 = rooted.nets.version
    rb, pb = rooted.move(), plain.move()
    torch.cuda.synchronize()
    assert rooted.nets.version == v0 + 1
    both = same & ((pb[1] - rb[1]).abs() <= 1e-5).all(dim=1)
    assert float(both.float().mean()) >= 0.7
    np.testing.assert_allclose(rb[2][both].cpu().numpy(), pb[2][both].cpu().numpy(), rtol=1e-4, atol=1e-6)
    rooted.env.close()
    plain.env.close()


def test_root_form_advances_running_statistics_like_the_plain_form():
    """track_running_stats with roots given as h's bottleneck values: the root expansion (f on the root, g for the A
    children, in the bottleneck form with weights composed with h's final convolution) leaves the same BatchNorm
    running statistics and call counts in the modules as the same expansion from the 256-channel latents."""
    import copy
    from muzero_amd.engine import BatchedMCTS
    from muzero_amd.towers import TowerNetworks
    from tools.bench_towers import build_nets
    A, G, S = 5, 6, 4
    g, f = build_nets((256, 6, 6), A)
    g2, f2 = copy.deepcopy(g), copy.deepcopy(f)
    torch.manual_seed(9)
    fin = torch.nn.Conv2d(3, 256, 1).cuda()
    rng = np.random.default_rng(11)
    root_t = torch.from_numpy(rng.standard_normal((G, 3, 6, 6)).astype(np.float32)).cuda()
    with torch.no_grad():
        lat = fin(root_t).reshape(G, -1).contiguous()
    eta = torch.from_numpy(rng.dirichlet(np.full(A, 0.3), size=G).astype(np.float32)).cuda()
    rand0 = torch.from_numpy(rng.integers(0, A, size=G).astype(np.int32)).cuda()
    to_play = torch.zeros(G, dtype=torch.int8, device="cuda")
    nets_r = TowerNetworks(g, f, track_running_stats=True, root_final=fin)
    nets_p = TowerNetworks(g2, f2, track_running_stats=True)
    rooted = BatchedMCTS(nets_r, G, S, 1, discount=0.997, noise_frac=0.25, keep_latents=False)
    plain = BatchedMCTS(nets_p, G, S, 1, discount=0.997, noise_frac=0.25, keep_latents=False)
    assert rooted.root_form_ok(root_t)
    rooted.begin(None, to_play, eta, rand0, root_t=root_t)
    plain.begin(lat, to_play, eta, rand0)
    nets_r.sync_running_stats()
    nets_p.sync_running_stats()
    torch.cuda.synchronize()
    bns = lambda *nets: [m for n in nets for m in n.modules() if isinstance(m, torch.nn.BatchNorm2d)]  # noqa: E731
    seen = 0
    for a, b in zip(bns(g, f), bns(g2, f2)):
        assert int(a.num_batches_tracked) == int(b.num_batches_tracked) == G
        for x, y in ((a.running_mean, b.running_mean), (a.running_var, b.running_var)):
            x, y = x.detach().double().cpu().numpy(), y.detach().double().cpu().numpy()
            np.testing.assert_allclose(x, y, rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(y).max())))
        seen += 1
    assert seen == 6   # three BatchNorm layers in g's block, three in f's


@pytest.mark.parametrize("rows", [5, 300, 4096])
def test_merge_conv_on_split_planes_equals_the_fp32_form(rows):
    """mz_merge_conv_forward_split (every float32 operand as two fp16 terms, three fp16 products per product in
    one float32 accumulator) against mz_merge_conv_forward (fp32 matrix instructions) on the shipped shape: the
    products Y agree to 1e-6 of their scale, for the expansion (all actions) and for one action per row, ragged row
    counts and gathered rows; and against a float64 evaluation both are equally close."""
    import ctypes as C
    from muzero_amd import _hip
    from muzero_amd._hip import ptr
    from muzero_amd.towers import TowerNetworks
    from tools.bench_towers import build_nets
    A = 5
    g, f = build_nets((256, 6, 6), A)
    nets = TowerNetworks(g, f)
    assert nets.fused_g and nets.split_merge_conv and "merge_t_perm" in nets.split3
    lib = _hip.lib()
    st = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator(device="cuda").manual_seed(rows)
    t = torch.randn(rows + 3, 108, device="cuda", generator=gen)
    t_rows = torch.randperm(rows + 3, device="cuda", generator=gen)[:rows].to(torch.int32)
    act = torch.randint(0, A, (rows,), device="cuda", generator=gen, dtype=torch.int32)
    c, hgt, wid = nets.shape
    for action, n_out in ((None, rows * A), (act, rows)):
        y32 = torch.zeros(n_out, hgt * wid, 32, device="cuda")
        y16 = torch.zeros_like(y32)
        _hip.check(lib.mz_merge_conv_forward(C.byref(nets.merge_t_perm), C.byref(nets.conv_g), ptr(nets.blob), ptr(t), 108,
                                             ptr(t_rows), ptr(nets.table_t_perm), ptr(action), A, rows, c, hgt * wid, ptr(y32), st),
                   "mz_merge_conv_forward")
        _hip.check(lib.mz_merge_conv_forward_split(C.byref(nets.merge_t_perm), ptr(nets.split3["merge_t_perm"]),
                                                   C.byref(nets.conv_g), ptr(nets.split3["conv_g"]), ptr(t), 108, ptr(t_rows),
                                                   ptr(nets.table_t_perm), ptr(action), A, rows, c, hgt * wid, ptr(y16), st),
                   "mz_merge_conv_forward_split")
        torch.cuda.synchronize()
        scale = float(y32.abs().max())
        assert scale > 0 and float((y16 - y32).abs().max()) <= 1e-6 * scale, (float((y16 - y32).abs().max()), scale)
    # float64 evaluation of the expansion for a few rows: z = W t, x = relu(z + table[a]), Y = x Wg (first block's columns)
    with torch.no_grad():
        blk = g.res_blocks[0]
        co = blk.conv1.out_channels
        sel = t_rows[:4].long()
        lat = nets_latent_from_t(nets, g, t[sel].double())
        base = lat @ g.fc_merge.weight[:, :nets.D].double().cuda().T
        x = torch.relu(base[:, None, :] + nets.table.double()[None, :, :]).view(4, A, c, hgt * wid)
        wg = torch.cat([blk.conv1.weight.double().cuda().permute(0, 2, 3, 1).reshape(co * 9, c),
                        blk.downsample[0].weight.double().cuda().reshape(co, c)])
        want = torch.einsum("racp,kc->rapk", x, wg)
    y16 = torch.zeros(rows * A, hgt * wid, 32, device="cuda")
    _hip.check(lib.mz_merge_conv_forward_split(C.byref(nets.merge_t_perm), ptr(nets.split3["merge_t_perm"]), C.byref(nets.conv_g),
                                               ptr(nets.split3["conv_g"]), ptr(t), 108, ptr(t_rows), ptr(nets.table_t_perm), None, A,
                                               rows, c, hgt * wid, ptr(y16), st), "mz_merge_conv_forward_split")
    torch.cuda.synchronize()
    got = y16.view(rows, A, hgt * wid, 32)[:4, :, :, :co * 10].double()
    assert float((got - want).abs().max()) <= 2e-6 * float(want.abs().max())


def nets_latent_from_t(nets, g, t64):
    """The latent a row of bottleneck values stands for: res_final of g applied to the block output (float64)."""
    c, hgt, wid = nets.shape
    fin = g.res_final
    w = fin.weight.double().cuda().reshape(c, -1)
    b = fin.bias.double().cuda() if fin.bias is not None else torch.zeros(c, dtype=torch.float64, device="cuda")
    tt = t64.view(t64.shape[0], -1, hgt * wid)
    return (torch.einsum("ck,rkp->rcp", w, tt) + b[None, :, None]).reshape(t64.shape[0], -1)


@pytest.mark.parametrize("G", [256, 4096])
def test_tower_search_in_the_benchmarked_form(G):
    """The tower search exactly as bench.py times it (bench.py: tower_search): roots given as the bottleneck values
    of h's final convolution (`root_t`), no stored latents, fc_merge fused with the first convolution on split planes,
    50 simulations, at 256 and at 4096 trees:
      (i)   launch by launch with tracing: the search's own network outputs replayed through the CPU tree oracle give
            EVERY tree bit for bit; the whole-search graph replay leaves the same pool;
      (ii)  against the separate fp32 kernels started from the full root latents: a tree either takes the same
            descents with the same rollouts, or its first difference is a PROVEN near tie (tests.gpu_util.explain_divergence);
      (iii) on 8 trees the prior / reward / rollout streams of the fused run equal the torch modules called one
            reference call at a time (src/search/expansion.py:25-33, src/search/simulation.py:31-44) on the latents
            its bottleneck values stand for, to 1e-5 of each output's scale."""
    from muzero_amd.engine import BatchedMCTS
    from muzero_amd.towers import TowerNetworks
    from tests.gpu_util import explain_divergence
    from tools.bench_towers import build_nets
    shape, A, S, depth = (256, 6, 6), 5, 50, 1
    D = 9216
    g, f = build_nets(shape, A)
    with torch.no_grad():
        for m in list(g.modules()) + list(f.modules()):
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
    torch.manual_seed(1)
    fin = torch.nn.Conv2d(3, shape[0], 1).cuda()
    fused = TowerNetworks(g, f, root_final=fin)
    plain = TowerNetworks(g, f, fuse_merge_conv=False)
    assert fused.fused_g and fused.has_root_form and not plain.fused_g and G >= fused.fuse_min_rows
    rng = np.random.default_rng(40 + G)
    root_t = torch.from_numpy(rng.standard_normal((G, 3, shape[1], shape[2])).astype(np.float32)).cuda()
    with torch.no_grad():
        lat = fin(root_t).reshape(G, D).contiguous()
    eta = rng.dirichlet(np.full(A, 0.3), size=G).astype(np.float32)
    rand = rng.integers(0, A, size=(G, S + 1)).astype(np.int32)
    to_play = np.zeros(G, np.int8)
    d_tp, d_eta, d_rand = [torch.from_numpy(x).cuda() for x in (to_play, eta, rand)]

    def searched(networks, rooted):
        mcts = BatchedMCTS(networks, G, S, depth, discount=0.998, noise_frac=0.25, keep_latents=False)
        kw = {"root_t": root_t} if rooted else {}
        assert mcts.t_pool is not None and not mcts.keep_latents and mcts.root_form_ok(kw.get("root_t")) == rooted
        trace = {}
        res = mcts.run(None if rooted else lat, d_tp, d_eta, d_rand, trace=trace, **kw)   # launch by launch (tracing)
        torch.cuda.synchronize()
        pool = mcts.pool
        prior_s = np.stack([p.cpu().numpy() for p in trace["prior"]], axis=1)
        rollout = np.stack([r.cpu().numpy() for r in trace["rollout"]], axis=1)
        leaf = np.stack([x.cpu().numpy() for x in trace["leaf"]], axis=1)
        child = np.stack([x.cpu().numpy() for x in trace["child"]], axis=1)
        reward_nodes = pool.reward.cpu().numpy()
        root_prior = torch.empty(G, A, device="cuda")
        rows0 = (torch.arange(G, device="cuda", dtype=torch.int32) * pool.cap).contiguous()
        if rooted:   # the root's pre-noise priors, evaluated as the search evaluated them
            networks.prediction(pool.latent, G, rows=rows0, softmax_times=1, policy=root_prior, t_in=(mcts.t_pool, rows0),
                                root_form=True)
        else:
            networks.prediction(lat, G, softmax_times=1, policy=root_prior)
        prior_stream = np.concatenate([root_prior.cpu().numpy()[:, None], prior_s], axis=1)
        reward_stream = np.stack([reward_nodes[:, 1 + e * A:1 + (e + 1) * A] for e in range(S + 1)], axis=1)
        ref = otree.run_batch(A, S, depth, to_play, prior_stream, reward_stream, rand, rollout, eta, 0.25, 0.998)
        assert np.array_equal(leaf, ref["leaf"]) and np.array_equal(child, ref["child"])
        for name in ("visit", "value_sum", "prior", "first_child", "parent"):
            assert np.array_equal(getattr(pool, name).cpu().numpy(), ref[name]), name
        assert np.array_equal(res.value.cpu().numpy(), ref["utility"])
        np.testing.assert_allclose(res.policy.cpu().numpy(), ref["policy"], rtol=1e-12)
        stats = {n: getattr(pool, n).clone() for n in ("visit", "value_sum", "prior", "reward", "parent", "first_child")}
        for _ in range(2):   # the first records the graph (the whole search when it starts from root_t), both replay it
            res2 = mcts.run(None if rooted else lat, d_tp, d_eta, d_rand, **kw)
            torch.cuda.synchronize()
            for n, want in stats.items():
                assert torch.equal(getattr(pool, n), want), f"graph replay: {n}"
            assert torch.equal(res2.policy32, res.policy32) and torch.equal(res2.value, res.value)
        assert mcts.replays == 2, "the recorded graph was not used"
        return dict(prior=prior_stream, reward=reward_stream, rollout=rollout, leaf=leaf, child=child, mcts=mcts)

    a = searched(fused, True)       # the benchmarked form
    b = searched(plain, False)      # separate fp32 kernels, full root latents
    assert not np.array_equal(a["prior"], b["prior"]), "the two runs went through the same kernels"

    # ---- (ii) every divergence a proven near tie
    t_pool, cap = a["mcts"].t_pool, a["mcts"].pool.cap

    def policy_of_child(gi, s):   # f's policy for the child simulation s started from, from the fused run's own values
        row = torch.tensor([gi * cap + int(a["child"][gi, s])], dtype=torch.int32, device="cuda")
        out = torch.empty(1, A, device="cuda")
        fused.prediction(a["mcts"].pool.latent, 1, rows=row, softmax_times=0, policy=out, t_in=(t_pool, row))
        return out[0].cpu().numpy()
    flips = {"select": 0, "rollout": 0}
    same = 0
    for gi in range(G):
        why = explain_divergence(a, b, gi, A, S, depth, to_play, eta, rand, 0.25, 0.998, policy_of_child)
        if why is None:
            same += 1
            for n in ("prior", "reward", "rollout"):
                np.testing.assert_allclose(a[n][gi], b[n][gi], rtol=1e-5, atol=max(1e-5 * float(np.abs(b[n][gi]).max()), TOL["atol"]),
                                           err_msg=f"{n} tree {gi}")
        else:
            flips[why[0]] += 1
    print(f"{G} trees: {same} identical descents, {flips['select']} proven selection near ties, "
          f"{flips['rollout']} proven greedy-action near ties")
    assert same + flips["select"] + flips["rollout"] == G and same >= 0.8 * G

    # ---- (iii) the torch modules, one reference call at a time, on the fused run's own latents
    gt, ft = g.cuda(), f.cuda()
    acts = torch.arange(A, device="cuda")

    def latent_of(gi, node):   # the latent a node's bottleneck values stand for (the root: h's final convolution of root_t)
        if node == 0:
            return lat[gi].reshape(1, *shape)
        t64 = t_pool[gi * cap + node].double().reshape(1, -1)
        return nets_latent_from_t(fused, g, t64).float().reshape(1, *shape)

    def close(got, want, what):
        want = want.detach().cpu().numpy()
        np.testing.assert_allclose(np.asarray(got).reshape(want.shape), want, rtol=1e-5,
                                   atol=max(1e-5 * float(np.abs(want).max()), TOL["atol"]), err_msg=what)
    near_ties = 0
    with torch.no_grad():
        for gi in range(0, G, G // 8):
            for s in list(range(0, S, 7)) + [S - 1]:
                x = latent_of(gi, int(a["leaf"][gi, s]))
                close(a["prior"][gi, s + 1], torch.softmax(ft(x)[0], dim=1)[0], f"prior tree {gi} sim {s}")
                _, rew = gt(x.repeat(A, 1, 1, 1), acts)   # ONE call: batch statistics over the A children
                close(a["reward"][gi, s + 1], rew[:, 0], f"reward tree {gi} sim {s}")
                cur = latent_of(gi, int(a["child"][gi, s]))
                pol = ft(cur)[0][0]
                top2 = pol.topk(2).values
                if float(top2[0] - top2[1]) <= 1e-5 * float(top2[0]):
                    near_ties += 1
                    continue
                cur, r = gt(cur, pol.argmax().reshape(1))
                v = ft(cur)[1]
                close(a["rollout"][gi, s], torch.stack([r[0, 0], v[0, 0]]), f"rollout tree {gi} sim {s}")
    assert near_ties <= 2
    g.cpu(), f.cpu()
