"""The reference's Python API on top of the GPU engine: MCTS.run(root) with the global RNGs
seeded as the reference's run was, strategy classes on hand-built Node trees, single-game
self-play and the learner, each against vectors captured from the reference."""
import os
import random

import numpy as np
import pytest
import torch

from tests.conftest import GOLDEN
from tests.test_oracle_golden import TREES, TREE_KEYS, _load_tiny_nets

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k", [TREE_KEYS[0], TREE_KEYS[7], TREE_KEYS[12], TREE_KEYS[25]])
def test_mcts_run_dropin_reproduces_reference_run(k):
    from muzero_amd.config import MCTSConfig
    from muzero_amd.search.factory import create_mcts
    from muzero_amd.search.nodes import Node
    z = TREES
    g, f, A = _load_tiny_nets(z, int(z[k + "net"]))
    S, depth = int(z[k + "S"]), int(z[k + "depth"])
    cfg = MCTSConfig(selection_strategy="puct", max_iterations=S, max_time=0, model_look_ahead=depth,
                     discount_factor=float(z[k + "discount"]), dirichlet_alpha=0.3, noise_frac=float(z[k + "noise_frac"]))
    mcts = create_mcts(g, f, torch.arange(A).long(), cfg)
    seed = int(z[k + "seed"])
    random.seed(seed)
    torch.manual_seed(seed)
    root = Node(latent_state=torch.from_numpy(z[k + "root_latent"]), to_play=int(z[k + "to_play"]))
    policy, utility = mcts.run(root)
    assert isinstance(policy, list) and len(policy) == A and isinstance(utility, float)
    assert root.visit_count == S  # tests/search/test_mcts.py:130 in the reference
    kids = list(root.children.values())
    assert len(kids) == A and all(c.parent is root and c.to_play == -root.to_play for c in kids)
    # RNG consumption: the next draws equal what the reference's run left behind
    random.seed(seed)
    torch.manual_seed(seed)
    [random.randrange(A) for _ in range(S + 1)]
    torch.distributions.Dirichlet(torch.full((A,), 0.3)).sample()
    want_next = (random.random(), torch.rand(1).item())
    random.seed(seed)
    torch.manual_seed(seed)
    mcts.run(Node(latent_state=torch.from_numpy(z[k + "root_latent"]), to_play=int(z[k + "to_play"])))
    assert (random.random(), torch.rand(1).item()) == want_next
    # root statistics: identical whenever no near-tie was flipped by fp32 rounding
    got_visits = [c.visit_count for c in kids]
    want_visits = z[k + "visit"][1:1 + A].tolist()
    np.testing.assert_allclose([float(c.policy_priority) for c in kids], z[k + "prior"][1:1 + A], rtol=1e-5)
    if got_visits == want_visits:
        np.testing.assert_allclose(policy, z[k + "policy"], rtol=1e-5)
        np.testing.assert_allclose(utility, float(z[k + "utility"]), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose([c.value_sum for c in kids], z[k + "value_sum"][1:1 + A], rtol=1e-4, atol=1e-6)
    else:
        pytest.xfail(f"fp32 rounding flipped a near-tie: visits {got_visits} vs {want_visits}")


def test_strategy_classes_known_answers():
    """tests/search/test_selection.py and test_backpropagation.py of the reference, run
    against the mirrored classes (which execute the kernels)."""
    from muzero_amd.search.backpropagation import Backpropagation
    from muzero_amd.search.nodes import Node
    from muzero_amd.search.selection import PUCT, UCT
    lat = torch.randn(1, 3, 3, 3)
    for strategy in (UCT(), PUCT()):
        root = Node(lat)
        assert strategy(root) is root
        root.visit_count = 16
        c1 = root.add_child(lat, torch.tensor(1))
        c1.visit_count, c1.value_sum = 10, 1
        c2 = root.add_child(lat, torch.tensor(2))
        c2.visit_count, c2.value_sum = 5, 10
        assert strategy(root) is c2
    for two_player, root_sum in ((False, 25), (True, 15)):
        root = Node(lat)
        root.visit_count, root.value_sum, root.to_play = 16, 20, 1
        c1 = root.add_child(lat, torch.tensor(1))
        c1.visit_count, c1.value_sum = 10, 5
        c2 = root.add_child(lat, torch.tensor(2))
        c2.visit_count, c2.value_sum = 5, 10
        c1.to_play = c2.to_play = 2 if two_player else 1
        Backpropagation(1.0)(c2, [5], 2 if two_player else 1)
        assert (c2.visit_count, c2.value_sum) == (6, 15)
        assert (root.visit_count, root.value_sum) == (17, root_sum)
        assert (c1.visit_count, c1.value_sum) == (10, 5)


def test_expand_and_simulate_shapes():
    """tests/search/test_expansion.py:17-45 and test_simulation.py:8-21 of the reference."""
    from muzero_amd.search.expansion import _transform_latent_state, expand_node
    from muzero_amd.search.nodes import Node
    from muzero_amd.search.simulation import MuZeroSimulation
    g, f, A = _load_tiny_nets(TREES, 0)
    lat = torch.randn(1, 2, 3, 3)
    root = Node(lat.clone())
    child = expand_node(root, torch.arange(A), g, f)
    assert len(root.children) == A and child in root.children.values()
    assert torch.equal(root.latent_state, lat)
    assert all(c.latent_state.shape == lat.shape for c in root.children.values())
    assert _transform_latent_state(5, lat).shape == (5, 2, 3, 3)
    assert abs(sum(c.policy_priority for c in root.children.values()) - 1.0) < 1e-5
    out = MuZeroSimulation(g, f, depth=3)(child)
    assert len(out) == 4 and all(isinstance(x, float) for x in out)
    with pytest.raises(ValueError):
        expand_node(root, torch.arange(0), g, f)


def test_uct_search_raises_like_reference():
    from muzero_amd.config import MCTSConfig
    from muzero_amd.search.factory import create_mcts
    from muzero_amd.search.nodes import Node
    g, f, A = _load_tiny_nets(TREES, 0)
    mcts = create_mcts(g, f, torch.arange(A), MCTSConfig(selection_strategy="uct", max_iterations=3, max_time=0))
    with pytest.raises((ValueError, ZeroDivisionError)):
        mcts.run(Node(torch.randn(1, 2, 3, 3)))


def _c4_nets_k2():
    from muzero_amd.config import DenseLayerConfig as D, DynamicsNetworkConfig, PredictionNetworkConfig, RepresentationNetworkConfig
    from muzero_amd.neural_networks import DynamicsNetwork, PredictionNetwork, RepresentationNetwork
    torch.manual_seed(0)
    h = RepresentationNetwork((2, 6, 7), (4, 6, 7), RepresentationNetworkConfig(history_length=2, downsample=[], res_net=[]))
    g = DynamicsNetwork((4, 6, 7), 7, DynamicsNetworkConfig(res_net=[], reward_net=[D(out_features=128)]))
    f = PredictionNetwork((4, 6, 7), 7, PredictionNetworkConfig(
        res_net=[], value_net=[D(out_features=256), D(out_features=128)], policy_net=[D(out_features=256), D(out_features=256)]))
    return h, g, f


def test_single_game_selfplay_matches_reference_episodes():
    from muzero_amd.config import MCTSConfig, TrainingDataGeneratorConfig
    from muzero_amd.training_data_generator import TrainingDataGenerator
    from oracle.c4 import C4Env
    z = np.load(os.path.join(GOLDEN, "selfplay_learner.npz"))
    h, g, f = _c4_nets_k2()
    cfg = TrainingDataGeneratorConfig(
        num_episodes=3, max_steps_per_episode=9, total_time=0, epsilon=0.6, epsilon_decay=1.0, min_epsilon=0.25,
        mcts=MCTSConfig(selection_strategy="puct", max_iterations=12, max_time=0, model_look_ahead=1,
                        discount_factor=0.998, dirichlet_alpha=0.3, noise_frac=0.25))
    random.seed(5)
    np.random.seed(5)
    torch.manual_seed(5)
    episodes = TrainingDataGenerator(C4Env(), h, g, f, cfg).generate_training_data(training_steps=0)
    assert len(episodes) == int(z["n_episodes"])
    for e, ep in enumerate(episodes):
        assert [c.best_action for c in ep.chunks] == z[f"ep{e}_action"].tolist()
        assert np.array_equal(np.stack([c.state.numpy()[0] for c in ep.chunks]), z[f"ep{e}_state"])
        assert [c.reward for c in ep.chunks] == z[f"ep{e}_reward"].tolist()
        np.testing.assert_allclose(np.stack([c.policy.numpy() for c in ep.chunks]), z[f"ep{e}_policy"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose([c.value for c in ep.chunks], z[f"ep{e}_value"], rtol=1e-5, atol=1e-7)


def test_learner_matches_reference_losses():
    from muzero_amd.config import TrainingConfig
    from muzero_amd.replay_buffer import ReplayBuffer
    from muzero_amd.training import NeuralNetworkManager
    from muzero_amd.training_data_generator import Chunk, Episode
    z = np.load(os.path.join(GOLDEN, "selfplay_learner.npz"))
    h, g, f = _c4_nets_k2()
    episodes = []
    for e in range(int(z["n_episodes"])):
        n = z[f"ep{e}_action"].shape[0]
        episodes.append(Episode(chunks=[Chunk(
            state=torch.from_numpy(z[f"ep{e}_state"][i:i + 1]), policy=torch.from_numpy(z[f"ep{e}_policy"][i]),
            reward=float(z[f"ep{e}_reward"][i]), value=float(z[f"ep{e}_value"][i]), best_action=int(z[f"ep{e}_action"][i]))
            for i in range(n)]))
    tcfg = TrainingConfig(
        learning_rate=0.01, weight_decay=1e-4, batch_size=3, replay_buffer_size=8, alpha=0.6, discount_factor=0.9,
        epochs=1, betas=(0.9, 0.999), roll_ahead=2, look_back=2, mini_batch_size=3, reward_coefficient=1.0,
        value_coefficient=0.5, policy_coefficient=0.25, min_learning_rate=0.0, total_training_steps=10,
        lr_schedule="cosine", optimizer="sgd", momentum=0.9)
    random.seed(9)
    np.random.seed(9)
    rb = ReplayBuffer(buffer_size=8, batch_size=3, alpha=0.6, beta=1.0)
    rb.add_episodes(episodes)
    mgr = NeuralNetworkManager(tcfg, h, g, f)
    last = mgr.train(rb)
    np.testing.assert_allclose(mgr.loss_history, z["learner_losses"], rtol=1e-5)
    np.testing.assert_allclose(last, float(z["learner_last"]), rtol=1e-5)
    np.testing.assert_allclose([p for _, p in rb.episode_buffer], z["learner_priorities"], rtol=1e-5)
    np.testing.assert_allclose(g.fc_merge.bias.detach().numpy(), z["learner_fc_merge_bias"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(f.policy_mlp[-2].weight.detach().numpy(), z["learner_policy_last_weight"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(h.final_conv.weight.detach().numpy(), z["learner_repr_weight"], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("device_rng", [False, True], ids=["host_streams", "device_streams"])
def test_batched_selfplay_engine_produces_valid_episodes(device_rng):
    """Lock-step self-play on the GPU env (random streams drawn on the host or on the device): every recorded episode is a legal game
    under the CPU rules, ends exactly where the rules say, and carries normalised policies."""
    from muzero_amd.config import MCTSConfig, TrainingDataGeneratorConfig
    from muzero_amd.selfplay import SelfPlayEngine
    from oracle.c4 import C4
    from tests.test_gpu_nets import _c4_nets
    h, g, f = _c4_nets(0)
    cfg = TrainingDataGeneratorConfig(
        num_episodes=1, max_steps_per_episode=42, total_time=0, epsilon=0.5, epsilon_decay=1.0, min_epsilon=0.25,
        mcts=MCTSConfig(selection_strategy="puct", max_iterations=8, max_time=0, model_look_ahead=1,
                        discount_factor=0.998, dirichlet_alpha=0.3, noise_frac=0.25))
    G = 256
    eng = SelfPlayEngine(h, g, f, G, cfg, seed=1, device_rng=device_rng)
    episodes = eng.generate(num_moves=30)
    assert len(episodes) >= G // 2
    for ep in episodes[:200]:
        game = C4()
        for i, c in enumerate(ep.chunks):
            assert np.array_equal(game.observe(), c.state.numpy()[0])
            assert abs(float(c.policy.sum()) - 1.0) < 1e-5
            r, d = game.step(c.best_action)
            assert r == c.reward
            assert d == (i == len(ep.chunks) - 1)


def test_selfplay_moves_replayed_from_one_graph_play_the_same_games():
    """graph_moves: after two ordinary moves every move (random draws, observation, history, h, search, action pick,
    environment step, restarts) is one graph replay.  With one seed the engine plays the games the ordinary engine
    plays: the graph-safe generator hands the recorded draws the offsets the ordinary calls would have used."""
    from muzero_amd.config import MCTSConfig, TrainingDataGeneratorConfig
    from muzero_amd.selfplay import SelfPlayEngine
    from tests.test_gpu_nets import _c4_nets
    h, g, f = _c4_nets(0)
    cfg = TrainingDataGeneratorConfig(
        num_episodes=1, max_steps_per_episode=42, total_time=0, epsilon=0.5, epsilon_decay=0.9, min_epsilon=0.25,
        mcts=MCTSConfig(selection_strategy="puct", max_iterations=8, max_time=0, model_look_ahead=1,
                        discount_factor=0.998, dirichlet_alpha=0.3, noise_frac=0.25))
    G = 64
    plain = SelfPlayEngine(h, g, f, G, cfg, seed=11, device_rng=True)
    replayed = SelfPlayEngine(h, g, f, G, cfg, seed=11, device_rng=True, graph_moves=True)
    for t in range(12):
        a = plain.move(training_steps=t)
        b = replayed.move(training_steps=t)   # epsilon changes from move to move: it is an input of the graph
        torch.cuda.synchronize()
        for x, y, name in zip(a, b, ("obs", "policy32", "value", "action", "reward", "ended")):
            assert torch.equal(x, y), (t, name)
    assert replayed._move_graph is not None
    want, got = plain.generate(num_moves=20), replayed.generate(num_moves=20)
    assert len(want) == len(got) > 0
    for ea, eb in zip(want, got):
        assert len(ea.chunks) == len(eb.chunks)
        assert all(torch.equal(ca.state, cb.state) and torch.equal(ca.policy, cb.policy) and ca.best_action == cb.best_action
                   for ca, cb in zip(ea.chunks, eb.chunks))


def test_actor_store_learner_loop_on_device():
    """Self-play -> device episode store -> batched learner step -> re-packed weights, all on
    the GPU: windows come out of the store in tensor form (z-targets by the HIP kernel), one
    gradient steps lower the loss on the same batch, and the search sees the new weights after
    PackedNetworks.refresh()."""
    from muzero_amd.config import MCTSConfig, TrainingDataGeneratorConfig
    from muzero_amd.episode_store import DeviceEpisodeStore
    from muzero_amd.learner import DataParallelLearner, batched_unroll_loss
    from muzero_amd.selfplay import SelfPlayEngine
    from muzero_amd.training import compute_z_targets_batch, z_targets_device
    from tests.test_gpu_nets import _c4_nets
    h, g, f = _c4_nets(0)
    h, g, f = h.cuda(), g.cuda(), f.cuda()
    cfg = TrainingDataGeneratorConfig(
        num_episodes=1, max_steps_per_episode=42, total_time=0, epsilon=1.0, epsilon_decay=1.0, min_epsilon=0.25,
        mcts=MCTSConfig(selection_strategy="puct", max_iterations=6, max_time=0, model_look_ahead=1,
                        discount_factor=0.998, dirichlet_alpha=0.3, noise_frac=0.25))
    G = 128
    eng = SelfPlayEngine(h, g, f, G, cfg, seed=3, device_rng=True)
    store = DeviceEpisodeStore(G, 42, (2, 6, 7), 7, capacity=512)
    for _ in range(45):
        store.record(*eng.move())
    assert store.size >= G
    gen = torch.Generator(device="cuda").manual_seed(1)
    batch, slots, k = store.sample_windows(64, 4, 5, h.history_length, 0.998, z_targets_device, generator=gen)
    # the device z-targets equal the host-array entry point of the same kernel
    sl = slots[:, None].cpu()
    steps = (k[:, None] + torch.arange(6, device="cuda")[None]).cpu()
    R = store.done["reward"].cpu()[sl, steps].double().numpy()
    V = store.done["value"].cpu()[sl, steps].numpy()
    np.testing.assert_array_equal(batch.z.cpu().numpy(), compute_z_targets_batch(R, V, 0.998).astype(np.float32))
    params = [p for n in (h, g, f) for p in n.parameters()]
    dp = DataParallelLearner(h, g, f, torch.optim.SGD(params, lr=1e-3), 5)
    before, _, _ = dp.step(batch)
    for _ in range(2):
        dp.step(batch)
    with torch.no_grad():
        after = float(batched_unroll_loss(h, g, f, batch, 5)[0] / batch.is_weights.sum())
    assert after < before  # plain gradient steps on one batch lower its loss
    # the actors pick the new weights up
    lat = torch.randn(G, 168, device="cuda")
    pol_old = torch.empty(G, 7, device="cuda")
    eng.nets.prediction(lat, G, softmax_times=1, policy=pol_old)
    eng.nets.refresh()
    pol_new = torch.empty(G, 7, device="cuda")
    eng.nets.prediction(lat, G, softmax_times=1, policy=pol_new)
    assert not torch.equal(pol_old, pol_new)
    with torch.no_grad():
        want = torch.softmax(f(lat.reshape(G, 4, 6, 7))[0], dim=1)
    torch.testing.assert_close(pol_new, want, rtol=1e-5, atol=2e-6)
    # prioritised replay on the device: the learner's per-window |loss| becomes the episodes' priority
    pbatch, pslots, _ = store.sample_prioritized(64, 4, 5, h.history_length, 0.998, z_targets_device, alpha=0.6,
                                                 beta=0.4, generator=gen)
    _, per_window, mine = dp.step(pbatch)
    store.update_priorities(pslots[mine.cuda()], per_window)
    touched = torch.zeros(store.cap + 1, dtype=torch.bool, device="cuda")
    touched[pslots] = True
    assert bool((store.priority[touched] != 1.0).all()) and bool((store.priority[: store.size][~touched[: store.size]] == 1.0).all())
    real = (pslots[mine.cuda()] < store.cap).cpu()
    assert float(store.max_priority) == max(1.0, float(per_window[real].max()) + 1e-6)
    assert float(pbatch.is_weights.max()) == 1.0 and len(pbatch) == 64
    assert bool(((pbatch.is_weights > 0).cpu() == (pslots < store.cap).cpu()).all())


def test_validation_rollout_matches_reference():
    """model_simulation (src/inference.py:20-87): a greedy episode with the real actions in the
    history ring, global RNGs seeded as the reference's runs."""
    from muzero_amd.config import DenseLayerConfig as D, DynamicsNetworkConfig, MCTSConfig, PredictionNetworkConfig, \
        RepresentationNetworkConfig
    from muzero_amd.inference import model_simulation
    from muzero_amd.neural_networks import DynamicsNetwork, PredictionNetwork, RepresentationNetwork
    from oracle.c4 import C4Env
    z = np.load(os.path.join(GOLDEN, "validation.npz"))
    torch.manual_seed(0)
    h = RepresentationNetwork((2, 6, 7), (4, 6, 7), RepresentationNetworkConfig(history_length=3, downsample=[], res_net=[]))
    g = DynamicsNetwork((4, 6, 7), 7, DynamicsNetworkConfig(res_net=[], reward_net=[D(out_features=128)]))
    f = PredictionNetwork((4, 6, 7), 7, PredictionNetworkConfig(
        res_net=[], value_net=[D(out_features=256), D(out_features=128)], policy_net=[D(out_features=256), D(out_features=256)]))
    cfg = MCTSConfig(selection_strategy="puct", max_iterations=16, max_time=0, model_look_ahead=1,
                     discount_factor=0.998, dirichlet_alpha=0.3, noise_frac=0.25)

    class Recording(C4Env):
        def __init__(self):
            super().__init__()
            self.actions, self.rewards = [], []

        def step(self, action):
            res = super().step(action)
            self.actions.append(int(action))
            self.rewards.append(float(res[1]))
            return res

    for ri in range(int(z["n_runs"])):
        env = Recording()
        seed = int(z[f"run{ri}_seed"])
        random.seed(seed)
        torch.manual_seed(seed)
        total = model_simulation(env, h, g, f, 30, cfg, device="cpu", human_mode=False)
        want = z[f"run{ri}_actions"].tolist()
        # a near-tie in the search policy may flip one arg-max; everything before it must agree
        same = next((i for i, (a, b) in enumerate(zip(env.actions, want)) if a != b), min(len(want), len(env.actions)))
        assert same >= len(want) - 2, f"run {ri}: actions diverge at move {same}: {env.actions} vs {want}"
        if env.actions == want:
            assert env.rewards == z[f"run{ri}_rewards"].tolist()
            assert total == float(z[f"run{ri}_total"])
