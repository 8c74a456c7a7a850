"""CPU checks of the host-side mirror of the reference API (no GPU): ring buffer / history
tensor and replay buffer against vectors captured from the reference, config schema against
the reference's own YAML files when they are present, oracle data restatements."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import lib as oracle_lib
from tests.conftest import GOLDEN

DATA = np.load(os.path.join(GOLDEN, "data_path.npz"))


def test_ring_buffer_history_matches_reference_vectors():
    from muzero_amd.ring_buffer import Frame, FrameRingBuffer, make_history_tensor
    hi = 0
    for K in (1, 3, 4):
        frames, actions = DATA[f"histseq_K{K}_frames"], DATA[f"histseq_K{K}_actions"]
        rb = FrameRingBuffer(size=K)
        for t in range(frames.shape[0]):
            fr = Frame(state=torch.from_numpy(frames[t]), action=int(actions[t]))
            if not rb.full():
                rb.fill(fr)
            else:
                rb.add(fr)
            assert int(DATA[f"hist{hi}_K"]) == K and int(DATA[f"hist{hi}_t"]) == t
            assert np.array_equal(make_history_tensor(rb).numpy(), DATA[f"hist{hi}_out"])
            hi += 1
    assert hi == int(DATA["n_hist"])


def test_oracle_history_and_ring_slots():
    """The C restatement reproduces the same tensors from slot-ordered frames."""
    hi = 0
    for K in (1, 3, 4):
        frames, actions = DATA[f"histseq_K{K}_frames"][:, 0], DATA[f"histseq_K{K}_actions"]
        T = frames.shape[0]
        slots = np.zeros(T, np.int32)
        oracle_lib().mzo_ring_slots(C.c_int32(K), C.c_int32(T), slots.ctypes.data_as(C.POINTER(C.c_int32)))
        ring_f = np.repeat(frames[0:1], K, axis=0).copy()
        ring_a = np.full(K, actions[0], np.int32)
        for t in range(T):
            if t > 0:
                ring_f[slots[t]] = frames[t]
                ring_a[slots[t]] = actions[t]
            Cc, H, W = frames.shape[1:]
            out = np.zeros((Cc * K + K, H, W), np.float32)
            oracle_lib().mzo_history(np.ascontiguousarray(ring_f).ctypes.data_as(C.POINTER(C.c_float)),
                                     ring_a.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int32(K), C.c_int32(Cc),
                                     C.c_int32(H), C.c_int32(W), out.ctypes.data_as(C.POINTER(C.c_float)))
            assert np.array_equal(out[None], DATA[f"hist{hi}_out"]), (K, t)
            hi += 1


def test_oracle_z_targets_match_reference_vectors():
    for i in range(int(DATA["n_z"])):
        w, gamma = int(DATA[f"z{i}_w"]), float(DATA[f"z{i}_gamma"])
        R, V = np.ascontiguousarray(DATA[f"z{i}_R"]), np.ascontiguousarray(DATA[f"z{i}_V"])
        z = np.zeros(w + 1)
        dp = C.POINTER(C.c_double)
        oracle_lib().mzo_z_targets(R.ctypes.data_as(dp), V.ctypes.data_as(dp), C.c_int32(w), C.c_double(gamma),
                                   z.ctypes.data_as(dp))
        assert np.array_equal(z, DATA[f"z{i}_z"])  # bit-exact float64


def test_replay_buffer_matches_reference_vectors():
    from muzero_amd.replay_buffer import ReplayBuffer
    from muzero_amd.training_data_generator import Chunk, Episode

    def fake_episode(n, base):
        return Episode(chunks=[Chunk(state=torch.zeros(1), policy=torch.zeros(1), reward=float(base + i), value=0.0,
                                     best_action=0) for i in range(n)])
    np.random.seed(1234)
    rb = ReplayBuffer(buffer_size=10, batch_size=4, alpha=0.6, beta=1.0)
    rb.add_episodes([fake_episode(3 + i, i) for i in range(12)])
    assert [ep.chunks[0].reward for ep, _ in rb.episode_buffer] == DATA["rp_kept_first_reward"].tolist()  # FIFO keeps newest
    _, i1, w1 = rb.sample_batch()
    rb.update_priorities(i1, [0.5, 2.0, 0.0, 3.5])
    _, i2, w2 = rb.sample_batch()
    rb.add_episodes([fake_episode(2, 100)])
    _, i3, w3 = rb.sample_batch()
    for got_i, got_w, k in ((i1, w1, 1), (i2, w2, 2), (i3, w3, 3)):
        assert np.array_equal(np.array(got_i), DATA[f"rp_idx{k}"])
        assert np.array_equal(np.array(got_w), DATA[f"rp_w{k}"])
    assert np.array_equal(np.array([p for _, p in rb.episode_buffer]), DATA["rp_prio_final"])
    assert rb.max_priority == float(DATA["rp_max_priority"])


def test_replay_buffer_constructor_errors_and_empty_sample():
    from muzero_amd.replay_buffer import ReplayBuffer
    for args in ((0, 1), (4, 0), (2, 3)):  # tests/test_replay_buffer.py:8-20 in the reference
        with pytest.raises(ValueError):
            ReplayBuffer(*args)
    assert ReplayBuffer(4, 2).sample_batch() == ([], [], [])


@pytest.mark.parametrize("name", ["config.yaml", "config_flappy_bird.yaml", "config_car_racing.yaml"])
def test_reference_yaml_files_load_with_the_mirrored_schema(name):
    path = os.path.join("/root/reference/src/config", name)
    if not os.path.exists(path):
        pytest.skip("reference checkout not present on this machine")
    from muzero_amd.config import load_config
    if name == "config_car_racing.yaml":
        # stale in the reference too: it lacks the required training.discount_factor
        # (src/config/config_loader.py:118), so the reference's own loader rejects it
        with pytest.raises(Exception, match="discount_factor"):
            load_config(path)
        return
    cfg = load_config(path)
    assert cfg.training_data_generator.mcts.max_iterations >= 0
    assert len(cfg.networks.latent_shape) == 3 and cfg.training_data_generator.min_epsilon == 0.25


def test_package_config_and_network_shapes():
    """Shape contract of the reference's network tests (tests/nerual_networks/test_nerual_network.py):
    g -> ([B,c,h,w], [B,1]); f -> ([B,A], [B,1]); h -> [B,c,h,w] from C*K+K input planes."""
    from muzero_amd.config import load_config
    from muzero_amd.neural_networks import DynamicsNetwork, PredictionNetwork, RepresentationNetwork
    cfg = load_config("config_c4_mlp.yaml")
    net_cfg = cfg.networks
    h = RepresentationNetwork((2, 6, 7), net_cfg.latent_shape, net_cfg.representation)
    g = DynamicsNetwork(net_cfg.latent_shape, 7, net_cfg.dynamics)
    f = PredictionNetwork(net_cfg.latent_shape, 7, net_cfg.prediction)
    n_reward_layers = len(net_cfg.dynamics.reward_net)
    DynamicsNetwork(net_cfg.latent_shape, 7, net_cfg.dynamics)
    assert len(net_cfg.dynamics.reward_net) == n_reward_layers  # caller's config list is not mutated
    s = h(torch.zeros(5, 3, 6, 7))
    assert s.shape == (5, 4, 6, 7)
    nxt, r = g(s, torch.zeros(5, dtype=torch.long))
    p, v = f(s)
    assert nxt.shape == s.shape and r.shape == (5, 1) and p.shape == (5, 7) and v.shape == (5, 1)
    torch.testing.assert_close(p.sum(dim=1), torch.ones(5))
    keys = set(g.state_dict()) | set(f.state_dict()) | set(h.state_dict())
    assert {"action_embedding.weight", "fc_merge.weight", "reward_mlp.0.weight", "value_mlp.4.bias",
            "policy_mlp.4.weight", "final_conv.weight"} <= keys


def test_node_defaults_and_add_child():
    """tests/search/test_node.py:6-41 in the reference."""
    from muzero_amd.search.nodes import Node
    lat = torch.randn(1, 3, 3)
    root = Node(lat)
    assert (root.visit_count, root.value_sum, root.reward, root.parent) == (0, 0.0, 0.0, None)
    a1, a2 = torch.tensor(1), torch.tensor(2)
    c1, c2 = root.add_child(lat, a1), root.add_child(lat, a2)
    assert list(root.children.values()) == [c1, c2] and c1.parent is root and root.children[a1] is c1
    root.to_play = 1
    assert root.add_child(lat, torch.tensor(3)).to_play == -1
