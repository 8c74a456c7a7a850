"""GPU parity of the tree kernels (select / expand-commit / root noise / backup / result)
against the CPU oracle and the reference-generated golden trees.  Bit-exact on every integer
and float64 statistic; the search-result softmax (device exp) within 1e-12 relative."""
import os

import numpy as np
import pytest
import torch

from oracle import tree as otree
from tests.conftest import GOLDEN
from tests.gpu_util import dev, random_streams, streams_to_device

pytestmark = pytest.mark.gpu


def _run_gpu(st, A, S, noise_frac, discount):
    from muzero_amd.engine import TreePool, run_recorded
    G = st["to_play"].shape[0]
    pool = TreePool(G, A, S, D=1, with_latent=False)
    d = streams_to_device(st)
    leaves, children = run_recorded(pool, d["to_play"], d["prior"], d["reward"], d["rand"], d["sim"], d["eta"],
                                    noise_frac, discount)
    policy, policy32, value, visits = pool.result()
    torch.cuda.synchronize()
    return pool, leaves.cpu().numpy().T, children.cpu().numpy().T, policy.cpu().numpy(), value.cpu().numpy(), \
        visits.cpu().numpy(), policy32.cpu().numpy()


def _compare(pool, leaves, children, policy, value, visits, ref):
    assert np.array_equal(leaves, ref["leaf"])
    assert np.array_equal(children, ref["child"])
    assert np.array_equal(pool.visit.cpu().numpy(), ref["visit"])
    assert np.array_equal(pool.value_sum.cpu().numpy(), ref["value_sum"])  # float64, bit-exact
    assert np.array_equal(pool.prior.cpu().numpy(), ref["prior"])
    assert np.array_equal(pool.parent.cpu().numpy(), ref["parent"])
    assert np.array_equal(pool.first_child.cpu().numpy(), ref["first_child"])
    assert np.array_equal(pool.to_play.cpu().numpy(), ref["to_play"])
    assert np.array_equal(pool.reward.cpu().numpy(), ref["reward"])
    assert np.array_equal(value, ref["utility"])  # one IEEE division
    np.testing.assert_allclose(policy, ref["policy"], rtol=1e-12)
    fc = ref["first_child"][:, 0]
    want_visits = np.stack([ref["visit"][g, fc[g]:fc[g] + visits.shape[1]] for g in range(visits.shape[0])])
    assert np.array_equal(visits, want_visits)


@pytest.mark.parametrize("A,S,depth,G,noise,disc", [
    (7, 50, 1, 4096, 0.25, 0.998),   # the benchmark shape
    (7, 50, 1, 1000, 0.0, 0.998),    # noise disabled, ragged last wave
    (2, 30, 2, 333, 0.25, 1.0),
    (5, 20, 3, 257, 0.25, 0.9),
    (18, 12, 1, 130, 0.25, 0.997),   # 32 lanes per tree
    (40, 6, 1, 65, 0.1, 0.997),      # one wavefront per tree
    (1, 9, 1, 70, 0.25, 0.99),       # single action: a chain
])
def test_recorded_streams_match_oracle(A, S, depth, G, noise, disc):
    st = random_streams(G, A, S, depth, seed=A * 100 + S)
    ref = otree.run_batch(A, S, depth, st["to_play"], st["prior"], st["reward"], st["rand"], st["sim"], st["eta"],
                          noise, disc)
    pool, leaves, children, policy, value, visits, _ = _run_gpu(st, A, S, noise, disc)
    _compare(pool, leaves, children, policy, value, visits, ref)
    assert np.all(pool.visit.cpu().numpy()[:, 0] == S)  # tests/search/test_mcts.py:130


def test_golden_reference_trees():
    """The trees captured from the reference itself, replayed through the kernels."""
    for fname, prefix, count in (("mcts_trees.npz", "c", "n_cases"), ("c4_mlp.npz", "t", "n_trees")):
        z = np.load(os.path.join(GOLDEN, fname))
        for i in range(int(z[count])):
            k = f"{prefix}{i}_"
            A, S, depth = int(z[k + "A"]), int(z[k + "S"]), int(z[k + "depth"])
            st = dict(prior=z[k + "prior_stream"][None], reward=z[k + "reward_stream"][None],
                      sim=z[k + "sim_stream"].astype(np.float32)[None], rand=z[k + "rand"][None],
                      eta=z[k + "eta"][None], to_play=np.array([int(z[k + "to_play"])], np.int8))
            pool, leaves, children, policy, value, visits, policy32 = _run_gpu(
                st, A, S, float(z[k + "noise_frac"]), float(z[k + "discount"]))
            assert np.array_equal(leaves[0], z[k + "leaf"]), k
            assert np.array_equal(children[0], z[k + "child"]), k
            assert np.array_equal(pool.visit.cpu().numpy()[0], z[k + "visit"]), k
            assert np.array_equal(pool.value_sum.cpu().numpy()[0], z[k + "value_sum"]), k
            assert np.array_equal(pool.prior.cpu().numpy()[0], z[k + "prior"]), k
            assert value[0] == float(z[k + "utility"]), k
            np.testing.assert_allclose(policy[0], z[k + "policy"], rtol=1e-12)
            np.testing.assert_allclose(policy32[0], z[k + "policy"].astype(np.float32), rtol=1e-6)


def test_selection_and_backup_known_answers():
    """tests/search/test_selection.py:16-41 and tests/search/test_backpropagation.py:15-81 on
    the kernels (bare trees: root children scored in float64)."""
    from muzero_amd.engine import TreePool
    pool = TreePool(3, 2, 4, D=1, with_latent=False)
    i32 = dict(dtype=torch.int32, device="cuda")
    pool.init(torch.tensor([1, 1, 1], dtype=torch.int8, device="cuda"))
    leaf = torch.zeros(3, **i32)
    pool.select(leaf, root_f32=False)
    assert leaf.tolist() == [0, 0, 0]  # empty root returns itself
    child = torch.zeros(3, **i32)
    z = torch.zeros((3, 2), dtype=torch.float32, device="cuda")
    pool.expand_commit(leaf, z, z, torch.zeros(3, **i32), child)
    pool.visit[:, 0] = 16
    pool.value_sum[:, 0] = 20.0
    pool.visit[:, 1], pool.value_sum[:, 1] = 10, 1.0
    pool.visit[:, 2], pool.value_sum[:, 2] = 5, 10.0
    depth = torch.zeros(3, **i32)
    pool.select(leaf, depth, root_f32=False)
    assert leaf.tolist() == [2, 2, 2] and depth.tolist() == [1, 1, 1]
    # backup: tree 0 single player (all to_play equal), tree 1 two player, tree 2 skipped
    pool.value_sum[:, 1] = 5.0
    pool.set_link((0, slice(1, 3)), to_play=1)
    pool.set_link((1, slice(1, 3)), to_play=2)
    start = torch.tensor([2, 2, -1], **i32)
    pool.backup(start, start.clamp(min=0), torch.full((3, 1), 5.0, dtype=torch.float32, device="cuda"), 1.0)
    torch.cuda.synchronize()
    assert pool.visit[0, :3].tolist() == [17, 10, 6] and pool.value_sum[0, :3].tolist() == [25.0, 5.0, 15.0]
    assert pool.visit[1, :3].tolist() == [17, 10, 6] and pool.value_sum[1, :3].tolist() == [15.0, 5.0, 15.0]
    assert pool.visit[2, :3].tolist() == [16, 10, 5]


def test_full_pool_and_table_overflow_are_reported():
    from muzero_amd.engine import TreePool
    A = 3
    pool = TreePool(2, A, 1, D=1, with_latent=False)  # cap = 1 + A + A: two expansions fit
    i32 = dict(dtype=torch.int32, device="cuda")
    pool.init(torch.zeros(2, dtype=torch.int8, device="cuda"))
    pr = torch.full((2, A), 1.0 / A, dtype=torch.float32, device="cuda")
    leaf, child = torch.zeros(2, **i32), torch.zeros(2, **i32)
    pool.expand_commit(leaf, pr, pr, torch.ones(2, **i32), child)
    assert child.tolist() == [2, 2]
    pool.expand_commit(child.clone(), pr, pr, torch.zeros(2, **i32), child)
    assert child.tolist() == [4, 4]
    nodes_before = pool.n_nodes.clone()
    pool.expand_commit(child.clone(), pr, pr, torch.zeros(2, **i32), child)  # pool is full now
    assert child.tolist() == [-1, -1] and torch.equal(pool.n_nodes, nodes_before)
    pool.visit[0, 0] = pool.tab_len + 5  # outside the PUCT tables
    pool.select(leaf)
    assert leaf[0].item() == -1 and leaf[1].item() >= 0


@pytest.mark.parametrize("A,S,G", [(7, 12, 300), (18, 6, 70), (40, 4, 65)])
def test_row_index_outputs(A, S, G):
    """mz_tree_select_rows / mz_tree_expand_commit_rows: the same leaves and children as the plain
    calls, plus pool rows tree * cap + node for the leaf, its future children and the drawn child."""
    from muzero_amd.engine import TreePool
    st = random_streams(G, A, S, 1, seed=A)
    d = streams_to_device(st)
    a, b = TreePool(G, A, S, D=1, with_latent=False), TreePool(G, A, S, D=1, with_latent=False)
    i32 = dict(dtype=torch.int32, device="cuda")
    leaf_a, leaf_b, child_a, child_b = (torch.zeros(G, **i32) for _ in range(4))
    row, child_row, kids = torch.zeros(G, **i32), torch.zeros(G, **i32), torch.zeros((G, A), **i32)
    base = torch.arange(G, **i32) * a.cap
    for p in (a, b):
        p.init(d["to_play"])
    a.expand_commit(leaf_a, d["prior"][0], d["reward"][0], d["rand"][0], child_a)
    b.expand_commit(leaf_b, d["prior"][0], d["reward"][0], d["rand"][0], child_b, out_child_row=child_row)
    assert torch.equal(child_a, child_b) and torch.equal(child_row, base + child_b)
    for s in range(S):
        a.select(leaf_a)
        n_before = b.n_nodes.clone()
        b.select(leaf_b, out_row=row, out_child_rows=kids)
        assert torch.equal(leaf_a, leaf_b) and torch.equal(row, base + leaf_b)
        assert torch.equal(kids, (base + n_before)[:, None] + torch.arange(A, **i32)[None, :])
        a.expand_commit(leaf_a, d["prior"][s + 1], d["reward"][s + 1], d["rand"][s + 1], child_a)
        b.expand_commit(leaf_b, d["prior"][s + 1], d["reward"][s + 1], d["rand"][s + 1], child_b, out_child_row=child_row)
        assert torch.equal(child_a, child_b) and torch.equal(child_row, base + child_b)
        a.backup(child_a, leaf_a, d["sim"][s], 0.997)
        b.backup(child_b, leaf_b, d["sim"][s], 0.997)
    assert torch.equal(a.visit, b.visit) and torch.equal(a.value_sum, b.value_sum)


@pytest.mark.parametrize("A,S,depth,G,noise,disc", [
    (7, 50, 1, 1000, 0.25, 0.998),
    (5, 50, 1, 256, 0.25, 0.998),    # the residual-tower configuration's tree shape
    (2, 30, 2, 333, 0.25, 1.0),
    (18, 12, 1, 130, 0.25, 0.997),   # 32 lanes per tree
    (40, 6, 1, 65, 0.1, 0.997),      # one wavefront per tree
    (1, 9, 1, 70, 0.25, 0.99),
])
def test_one_launch_per_step_matches_oracle(A, S, depth, G, noise, disc):
    """mz_tree_commit_backup_select (child records + backup of a step and the next step's descent in one launch):
    the same bit-exact trees as the three kernels, and the pool rows it reports for the leaf, its children-to-be
    and the drawn child."""
    from muzero_amd.engine import TreePool, run_recorded
    st = random_streams(G, A, S, depth, seed=A * 100 + S + 1)
    ref = otree.run_batch(A, S, depth, st["to_play"], st["prior"], st["reward"], st["rand"], st["sim"], st["eta"],
                          noise, disc)
    pool = TreePool(G, A, S, D=1, with_latent=False)
    d = streams_to_device(st)
    leaves, children, (rows, crows, drawn) = run_recorded(pool, d["to_play"], d["prior"], d["reward"], d["rand"], d["sim"],
                                                          d["eta"], noise, disc, fused_steps=True)
    policy, _, value, visits = pool.result()
    torch.cuda.synchronize()
    _compare(pool, leaves.cpu().numpy().T, children.cpu().numpy().T, policy.cpu().numpy(), value.cpu().numpy(),
             visits.cpu().numpy(), ref)
    base = (np.arange(G) * pool.cap)[None, :]
    assert np.array_equal(rows.cpu().numpy(), base + ref["leaf"].T)
    assert np.array_equal(drawn.cpu().numpy(), base + ref["child"].T)
    first = 1 + A * (1 + np.arange(S))[:, None, None]        # lock step: step s creates nodes 1 + A (s + 1) ...
    assert np.array_equal(crows.cpu().numpy(), base[:, :, None] + first + np.arange(A)[None, None, :])
