"""End-to-end GPU search (tree kernels + fused network kernels, lock step over G trees)
checked three ways:
  1. plumbing: replaying the network outputs the GPU itself produced through the CPU oracle
     must reproduce every GPU tree bit for bit (visit counts, value sums, leaves, children);
  2. numerics: those network outputs match torch fp32 on the same latents to 1e-5;
  3. against the per-node CPU port with live torch networks: trees whose every network
     output pair agrees in argmax terms must match exactly; the rest are counted and bounded.
"""
import numpy as np
import pytest
import torch

from oracle import pyref, tree as otree
from tests.test_gpu_nets import _c4_nets, _custom_nets

pytestmark = pytest.mark.gpu


def _inputs(G, A, S, D, seed):
    rng = np.random.default_rng(seed)
    lat = rng.standard_normal((G, D)).astype(np.float32)
    eta = rng.gamma(0.3, size=(G, A)).astype(np.float32) + 1e-6
    eta /= eta.sum(-1, keepdims=True)
    rand = rng.integers(0, A, size=(G, S + 1)).astype(np.int32)
    to_play = rng.choice(np.array([1, -1], np.int8), size=G)
    return lat, eta.astype(np.float32), rand, to_play


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "lockstep"])
@pytest.mark.parametrize("which,G,S,depth,noise", [("c4", 64, 50, 1, 0.25), ("c4", 9, 20, 2, 0.0), ("tiny", 40, 30, 3, 0.25)])
def test_search_replays_bit_exact_through_oracle(which, G, S, depth, noise, fused):
    from muzero_amd.engine import BatchedMCTS, PackedNetworks
    if which == "c4":
        _, g, f = _c4_nets(0)
        latent_shape = (4, 6, 7)
    else:
        g, f = _custom_nets((2, 3, 3), 5, [16], [16], [16])
        latent_shape = (2, 3, 3)
    A, D = g.num_actions, int(np.prod(latent_shape))
    nets = PackedNetworks(g, f)
    lat, eta, rand, to_play = _inputs(G, A, S, D, seed=G + S)
    mcts = BatchedMCTS(nets, G, S, depth, discount=0.998, noise_frac=noise)
    trace = {}
    res = mcts.run(torch.from_numpy(lat).cuda(), torch.from_numpy(to_play).cuda(), torch.from_numpy(eta).cuda(),
                   torch.from_numpy(rand).cuda(), trace=trace, fused=fused)
    torch.cuda.synchronize()
    pool = mcts.pool
    # --- 1. replay the GPU's own network outputs through the oracle
    fc0 = 1
    prior0 = pool.prior[:, fc0:fc0 + A].cpu().numpy()  # root priors after noise: recover pre-noise below
    prior_s = np.stack([p.cpu().numpy() for p in trace["prior"]], axis=1)           # [G, S, A]
    rollout = np.stack([r.cpu().numpy() for r in trace["rollout"]], axis=1)         # [G, S, depth+1]
    # root expansion priors/rewards: read the recorded node fields (reward is stored verbatim)
    reward_nodes = pool.reward.cpu().numpy()
    root_reward = reward_nodes[:, 1:1 + A]
    # pre-noise root priors: re-evaluate f on the root latent with the kernel itself
    root_prior = torch.empty(G, A, device="cuda")
    nets.prediction(torch.from_numpy(lat).cuda(), G, softmax_times=1, policy=root_prior)
    prior_stream = np.concatenate([root_prior.cpu().numpy()[:, None], prior_s], axis=1)
    child = np.stack([c.cpu().numpy() for c in trace["child"]], axis=1)             # [G, S]
    leaf = np.stack([l.cpu().numpy() for l in trace["leaf"]], axis=1)
    first_child = pool.first_child.cpu().numpy()
    reward_stream = np.zeros((G, S + 1, A), np.float32)
    reward_stream[:, 0] = root_reward
    for s in range(S):
        base = 1 + A + s * A
        reward_stream[:, s + 1] = reward_nodes[:, base:base + A]
    ref = otree.run_batch(A, S, depth, to_play, prior_stream, reward_stream, rand, rollout, eta, noise, 0.998)
    assert np.array_equal(leaf, ref["leaf"])
    assert np.array_equal(child, ref["child"])
    assert np.array_equal(pool.visit.cpu().numpy(), ref["visit"])
    assert np.array_equal(pool.value_sum.cpu().numpy(), ref["value_sum"])
    assert np.array_equal(pool.prior.cpu().numpy(), ref["prior"])
    assert np.array_equal(first_child, ref["first_child"])
    np.testing.assert_allclose(res.policy.cpu().numpy(), ref["policy"], rtol=1e-12)
    assert np.array_equal(res.value.cpu().numpy(), ref["utility"])
    assert (res.visits.sum(dim=1) == S).all()

    # --- 2. network outputs used by the search vs torch fp32 on the same latents
    lat_pool = pool.latent.cpu()
    with torch.no_grad():
        s_idx = S - 1
        leaf_lat = lat_pool[torch.arange(G), torch.from_numpy(leaf[:, s_idx]).long()].reshape(G, *latent_shape)
        want_prior = torch.softmax(f(leaf_lat)[0], dim=1).numpy()
        np.testing.assert_allclose(prior_s[:, s_idx], want_prior, rtol=1e-5, atol=2e-6)
        base = 1 + A + s_idx * A
        want_next, want_r = g(leaf_lat.repeat_interleave(A, 0), torch.arange(A).repeat(G))
        np.testing.assert_allclose(lat_pool[:, base:base + A].reshape(G * A, D).numpy(),
                                   want_next.reshape(G * A, D).numpy(), rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(reward_nodes[:, base:base + A].reshape(-1), want_r[:, 0].numpy(), rtol=1e-5, atol=2e-6)

        # the rollout stream of the same simulation, directly (in the fused kernel it comes through the
        # folded value / reward heads reading relu(z + table[action]) on the fly): greedy steps from
        # the drawn child's latent with the torch modules
        cur = lat_pool[torch.arange(G), torch.from_numpy(child[:, s_idx]).long()].reshape(G, *latent_shape)
        want_roll = []
        tie = torch.zeros(G, dtype=torch.bool)   # trees whose greedy action is decided by less than 1e-5
        for _ in range(depth):
            pol = f(cur)[0]
            top2 = pol.topk(2, dim=1).values
            tie |= (top2[:, 0] - top2[:, 1]) <= 1e-5 * top2[:, 0]
            cur, r = g(cur, pol.argmax(dim=1))
            want_roll.append(r[:, 0])
        want_roll.append(f(cur)[1][:, 0])
        want_roll = torch.stack(want_roll, dim=1).numpy()
        keep = ~tie.numpy()
        assert keep.sum() >= G - 2
        np.testing.assert_allclose(rollout[keep, s_idx], want_roll[keep], rtol=1e-5, atol=2e-6)

    # --- 3. the CPU port with live torch networks: identical trees, except where a decision was a
    # PROVEN near tie (the two candidates closer than the measured fp32 discrepancy between the MFMA
    # kernels and the CPU GEMMs, itself bounded by the 1e-5 network tolerance)
    torch.set_num_threads(1)
    same = flips_select = flips_rollout = 0
    n_cmp = min(G, 12)
    for gi in range(n_cmp):
        def port(n_sims, tr=None):
            ev = pyref.TorchEvaluator(g, f, torch.arange(A), depth)
            it = iter(rand[gi].tolist())
            with torch.no_grad():
                return pyref.run_mcts(torch.from_numpy(lat[gi]).reshape(1, *latent_shape), int(to_play[gi]), ev, A, n_sims,
                                      discount=0.998, noise_frac=noise, eta=eta[gi], rand_index=lambda n: next(it), trace=tr)
        tr = []
        policy, utility, nodes = port(S, tr)
        arr = pyref.pool_arrays(nodes, A)
        if np.array_equal(arr["visit"], ref["visit"][gi]):
            same += 1
            np.testing.assert_allclose(arr["value_sum"], ref["value_sum"][gi], rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(np.array(policy), ref["policy"][gi], rtol=1e-5)
            np.testing.assert_allclose(utility, ref["utility"][gi], rtol=1e-5, atol=1e-7)
            continue
        # first simulation whose rollout or selected leaf differs
        scale = max(1.0, float(np.abs(rollout[gi]).max()))
        s_star = kind = None
        for s_i in range(S):
            if tr[s_i][0] != leaf[gi, s_i]:
                s_star, kind = s_i, "select"
                break
            if np.abs(np.array(tr[s_i][3]) - rollout[gi, s_i]).max() > 1e-4 * scale:
                s_star, kind = s_i, "rollout"
                break
        assert s_star is not None, "trees differ although every simulation selected the same leaf with the same rollout"
        _, _, nodes_b = port(s_star)   # the port's tree right before the diverging simulation
        if kind == "rollout":
            # the greedy rollout took another action: the policy's two largest entries must be a near tie
            _, _, nodes_c = port(s_star + 1)
            cur = nodes_c[tr[s_star][1]].latent
            near = False
            with torch.no_grad():
                for _ in range(depth):
                    pol = f(cur)[0][0]
                    top2 = pol.topk(2).values
                    near |= bool((top2[0] - top2[1]) <= 1e-5 * top2[0])
                    cur, _ = g(cur, pol.argmax().reshape(1))
            assert near, f"tree {gi}: rollout of simulation {s_star} differs from the CPU port without a near-tie argmax"
            flips_rollout += 1
            continue
        # selection flip: descend the port's tree; at the first level where the GPU's path leaves the
        # port's choice, the two candidates' scores must be closer than the discrepancy between the
        # GPU's and the port's scores of that level (the GPU tree before this simulation = the oracle
        # replay of the GPU's own streams, proven bit-identical above)
        gpu_b = otree.run_batch(A, s_star, depth, to_play[gi:gi + 1], prior_stream[gi:gi + 1, :s_star + 1],
                                reward_stream[gi:gi + 1, :s_star + 1], rand[gi:gi + 1, :s_star + 1],
                                rollout[gi:gi + 1, :s_star], eta[gi:gi + 1], noise, 0.998)
        path = [int(leaf[gi, s_star])]
        par = pool.parent[gi].cpu().numpy()
        while path[-1] != 0:
            path.append(int(par[path[-1]]))
        path.reverse()    # root .. GPU leaf (all of them exist in both trees: same expansions so far)
        node, proven = nodes_b[0], False
        for nxt in path[1:]:
            sq = np.sqrt(node.visit)
            cpu_scores = np.array([float(pyref._puct_score(c, sq)) for c in node.children])
            gv, gw, gp = gpu_b["visit"][0], gpu_b["value_sum"][0], gpu_b["prior"][0]
            g_scores = []
            for c in node.children:
                shadow = pyref.PNode(None, None, 0, c.index)
                shadow.visit, shadow.value_sum = int(gv[c.index]), float(gw[c.index])
                shadow.prior = np.float32(gp[c.index]) if node.index == 0 else float(gp[c.index])
                g_scores.append(float(pyref._puct_score(shadow, np.sqrt(float(gv[node.index])))))
            g_scores = np.array(g_scores)
            best = int(np.argmax(cpu_scores))
            took = [c.index for c in node.children].index(nxt)
            if took != best:
                disc = float(np.abs(cpu_scores - g_scores).max())
                gap = float(cpu_scores[best] - cpu_scores[took])
                mag = max(1.0, float(np.abs(cpu_scores).max()))
                assert disc <= 1e-4 * mag, f"tree {gi}: scores of GPU and CPU port differ by {disc}"
                assert gap <= 2 * disc + 1e-12, (f"tree {gi}, simulation {s_star}: selection differs from the CPU port "
                                                 f"by a score gap of {gap}, discrepancy only {disc}")
                proven = True
                break
            node = node.children[took]
        assert proven
        flips_select += 1
    print(f"CPU port: {same}/{n_cmp} trees identical, {flips_select} proven selection near-tie flips, "
          f"{flips_rollout} proven rollout-argmax near-tie flips")
    assert same + flips_select + flips_rollout == n_cmp


@pytest.mark.parametrize("which,G,S,depth", [("c4", 100, 50, 1), ("c4", 16, 5, 0), ("tiny", 33, 25, 2), ("odd", 21, 12, 1),
                                             ("wide", 19, 6, 2), ("mid", 40, 20, 1), ("ragged", 35, 12, 1)])
def test_fused_kernel_equals_lockstep_kernels_bit_for_bit(which, G, S, depth):
    """One launch for the whole search vs the per-stage kernels: identical pools (every
    statistic and every latent), identical results."""
    from muzero_amd.engine import BatchedMCTS, PackedNetworks
    if which == "c4":
        _, g, f = _c4_nets(2)
        latent_shape = (4, 6, 7)
    elif which == "tiny":
        g, f = _custom_nets((2, 3, 3), 5, [16], [16], [16])
        latent_shape = (2, 3, 3)
    elif which == "odd":  # reward head without hidden layer, deep policy, mixed activations, A = 3
        g, f = _custom_nets((1, 1, 8), 3, [], [33], [17, 9, 40], ("tanh", "sigmoid", "relu"))
        latent_shape = (1, 1, 8)
    elif which == "mid":  # the sequential-program kernel's second instantiation
        g, f = _custom_nets((4, 4, 4), 5, [128], [128, 128], [128, 128], ("relu",))
        latent_shape = (4, 4, 4)
    elif which == "ragged":  # sequential-program kernel, a latent width that is no multiple of 4 (element-wise row accesses)
        g, f = _custom_nets((1, 10, 17), 6, [128], [256, 128], [256, 256], ("relu",))
        latent_shape = (1, 10, 17)
    else:  # 18 actions (32 lanes per tree, three reward chunks), two hidden reward layers
        g, f = _custom_nets((2, 4, 4), 18, [24, 12], [8], [24], ("relu",))
        latent_shape = (2, 4, 4)
    A, D = g.num_actions, int(np.prod(latent_shape))
    nets = PackedNetworks(g, f)
    lat, eta, rand, to_play = _inputs(G, A, S, D, seed=7 * G + S)
    args = (torch.from_numpy(lat).cuda(), torch.from_numpy(to_play).cuda(), torch.from_numpy(eta).cuda(),
            torch.from_numpy(rand).cuda())
    a = BatchedMCTS(nets, G, S, depth, discount=0.997, noise_frac=0.25)
    b = BatchedMCTS(nets, G, S, depth, discount=0.997, noise_frac=0.25)
    # without the merge-row workspace the single launch is always the general kernel; with it, searches of the
    # covered shape take the sequential-program kernel, which then writes the node latents too
    c = BatchedMCTS(nets, G, S, depth, discount=0.997, noise_frac=0.25, merge_row_workspace=False)
    assert a.z_pool is not None and c.z_pool is None
    ra = a.run(*args, fused=True)
    rb = b.run(*args, fused=False)
    rc = c.run(*args, fused=True)
    torch.cuda.synchronize()
    for m, r in ((a, ra), (c, rc)):
        for name in ("visit", "value_sum", "prior", "reward", "parent", "first_child", "to_play", "n_nodes", "latent"):
            assert torch.equal(getattr(m.pool, name), getattr(b.pool, name)), name
        assert torch.equal(r.policy, rb.policy) and torch.equal(r.value, rb.value) and torch.equal(r.visits, rb.visits)
        assert torch.equal(r.policy32, rb.policy32)


def _replay_through_oracle(mcts, nets, res, trace, lat, eta, rand, to_play, depth, noise, discount):
    """Feeds the network outputs the GPU search itself used back through the CPU oracle and
    demands identical trees (every node of every tree)."""
    pool, G, A, S = mcts.pool, mcts.G, mcts.A, mcts.S
    prior_s = np.stack([p.cpu().numpy() for p in trace["prior"]], axis=1)
    rollout = np.stack([r.cpu().numpy() for r in trace["rollout"]], axis=1)
    reward_nodes = pool.reward.cpu().numpy()
    root_prior = torch.empty(G, A, device="cuda")
    nets.prediction(torch.from_numpy(lat).cuda(), G, softmax_times=1, policy=root_prior)
    prior_stream = np.concatenate([root_prior.cpu().numpy()[:, None], prior_s], axis=1)
    reward_stream = np.zeros((G, S + 1, A), np.float32)
    for e in range(S + 1):
        base = 1 + e * A
        reward_stream[:, e] = reward_nodes[:, base:base + A]
    ref = otree.run_batch(A, S, depth, to_play, prior_stream, reward_stream, rand, rollout, eta, noise, discount)
    leaf = np.stack([l.cpu().numpy() for l in trace["leaf"]], axis=1)
    child = np.stack([c.cpu().numpy() for c in trace["child"]], axis=1)
    assert np.array_equal(leaf, ref["leaf"]) and np.array_equal(child, ref["child"])
    for name in ("visit", "value_sum", "prior", "first_child", "parent", "to_play"):
        assert np.array_equal(getattr(pool, name).cpu().numpy(), ref[name]), name
    np.testing.assert_allclose(res.policy.cpu().numpy(), ref["policy"], rtol=1e-12)
    assert np.array_equal(res.value.cpu().numpy(), ref["utility"])
    return ref


def test_full_size_search_replays_and_is_deterministic():
    """The benchmark configuration itself (4096 games x 50 simulations, c4_mlp): every one of the
    4096 trees replays bit for bit through the oracle, the structural invariants of a finished
    search hold, and a second launch reproduces the first bit for bit."""
    from muzero_amd.engine import BatchedMCTS, PackedNetworks
    _, g, f = _c4_nets(0)
    G, S, depth, A, D = 4096, 50, 1, 7, 168
    nets = PackedNetworks(g, f)
    lat, eta, rand, to_play = _inputs(G, A, S, D, seed=99)
    dev = [torch.from_numpy(x).cuda() for x in (lat, to_play, eta, rand)]
    a = BatchedMCTS(nets, G, S, depth, discount=0.998, noise_frac=0.25)
    trace = {}
    res = a.run(*dev, trace=trace)
    torch.cuda.synchronize()
    _replay_through_oracle(a, nets, res, trace, lat, eta, rand, to_play, depth, 0.25, 0.998)
    visit = a.pool.visit.cpu().numpy()
    assert (visit[:, 0] == S).all() and (visit[:, 1:1 + A].sum(1) == S).all()
    assert (res.visits.cpu().numpy() == visit[:, 1:1 + A]).all()
    assert (a.pool.n_nodes.cpu().numpy() == 1 + (S + 1) * A).all()
    np.testing.assert_allclose(res.policy.sum(1).cpu().numpy(), 1.0, rtol=1e-12)
    parent = a.pool.parent.cpu().numpy()
    fc = a.pool.first_child.cpu().numpy()
    # every non-root node's parent points at it through first_child .. first_child + A
    idx = np.arange(1, parent.shape[1])
    pfc = np.take_along_axis(fc, parent[:, 1:], axis=1)
    assert ((idx[None] >= pfc) & (idx[None] < pfc + A)).all()
    b = BatchedMCTS(nets, G, S, depth, discount=0.998, noise_frac=0.25)
    rb = b.run(*dev)
    torch.cuda.synchronize()
    for name in ("visit", "value_sum", "prior", "reward", "latent"):
        assert torch.equal(getattr(a.pool, name), getattr(b.pool, name)), name
    assert torch.equal(res.policy32, rb.policy32) and torch.equal(res.value, rb.value)


@pytest.mark.parametrize("G,S,depth", [(1, 1, 1), (17, 2, 0), (31, 3, 2), (4100, 4, 1)])
def test_ragged_and_minimal_batches(G, S, depth):
    """Batches that do not fill a 16-tree workgroup, a single tree, a single simulation: fused
    and per-stage kernels agree bit for bit and both replay through the oracle."""
    from muzero_amd.engine import BatchedMCTS, PackedNetworks
    _, g, f = _c4_nets(1)
    A, D = 7, 168
    nets = PackedNetworks(g, f)
    lat, eta, rand, to_play = _inputs(G, A, S, D, seed=3 * G + S)
    dev = [torch.from_numpy(x).cuda() for x in (lat, to_play, eta, rand)]
    pools = []
    for fused in (True, False):
        m = BatchedMCTS(nets, G, S, depth, discount=0.99, noise_frac=0.25)
        trace = {}
        res = m.run(*dev, trace=trace, fused=fused)
        torch.cuda.synchronize()
        _replay_through_oracle(m, nets, res, trace, lat, eta, rand, to_play, depth, 0.25, 0.99)
        pools.append((m, res))
    for name in ("visit", "value_sum", "prior", "reward", "latent"):
        assert torch.equal(getattr(pools[0][0].pool, name), getattr(pools[1][0].pool, name)), name
    assert torch.equal(pools[0][1].policy32, pools[1][1].policy32)


@pytest.mark.parametrize("which,G,S,depth", [("c4", 100, 50, 1), ("c4", 4100, 6, 2), ("tiny", 33, 25, 2), ("odd", 21, 12, 1),
                                             ("wide", 19, 6, 0), ("mid", 40, 20, 1), ("ragged", 35, 12, 1)])
def test_implicit_child_latents_leave_identical_trees(which, G, S, depth):
    """keep_latents=False (one merge pre-activation row per expansion instead of A child latents; a leaf's
    latent is formed from its parent's row when it is expanded, src/search/expansion.py:25-33): every tree
    statistic and every result equals the run that stores all latents, bit for bit, and the search's own
    network outputs replay through the oracle."""
    from muzero_amd.engine import BatchedMCTS, PackedNetworks
    if which == "c4":
        _, g, f = _c4_nets(2)
        latent_shape = (4, 6, 7)
    elif which == "tiny":
        g, f = _custom_nets((2, 3, 3), 5, [16], [16], [16])
        latent_shape = (2, 3, 3)
    elif which == "odd":
        g, f = _custom_nets((1, 1, 8), 3, [], [33], [17, 9, 40], ("tanh", "sigmoid", "relu"))
        latent_shape = (1, 1, 8)
    elif which == "mid":   # the sequential-program kernel's second instantiation: 64-wide latents, 128-wide heads
        g, f = _custom_nets((4, 4, 4), 5, [128], [128, 128], [128, 128], ("relu",))
        latent_shape = (4, 4, 4)
    elif which == "ragged":   # the first instantiation with a 170-wide latent
        g, f = _custom_nets((1, 10, 17), 6, [128], [256, 128], [256, 256], ("relu",))
        latent_shape = (1, 10, 17)
    else:
        g, f = _custom_nets((2, 4, 4), 18, [24, 12], [8], [24], ("relu",))
        latent_shape = (2, 4, 4)
    A, D = g.num_actions, int(np.prod(latent_shape))
    nets = PackedNetworks(g, f)
    lat, eta, rand, to_play = _inputs(G, A, S, D, seed=11 * G + S)
    args = [torch.from_numpy(x).cuda() for x in (lat, to_play, eta, rand)]
    a = BatchedMCTS(nets, G, S, depth, discount=0.997, noise_frac=0.25)
    b = BatchedMCTS(nets, G, S, depth, discount=0.997, noise_frac=0.25, keep_latents=False)
    assert b.implicit and b.pool.latent is None and not a.implicit
    ra = a.run(*args)
    trace = {}
    rb = b.run(*args, trace=trace)
    torch.cuda.synchronize()
    for name in ("visit", "value_sum", "prior", "reward", "parent", "first_child", "to_play", "n_nodes"):
        assert torch.equal(getattr(a.pool, name), getattr(b.pool, name)), name
    assert torch.equal(ra.policy, rb.policy) and torch.equal(ra.value, rb.value) and torch.equal(ra.visits, rb.visits)
    assert torch.equal(ra.policy32, rb.policy32)
    _replay_through_oracle(b, nets, rb, trace, lat, eta, rand, to_play, depth, 0.25, 0.997)
    # the rows it kept are the merge pre-activations the explicit run formed its children from
    zp = b.z_pool.cpu()
    lat_a = a.pool.latent.cpu()
    with torch.no_grad():
        table = g.fc_merge.weight[:, D:] @ g.action_embedding.weight.t() + g.fc_merge.bias[:, None]   # [D, A]
    for e in (0, S):
        kids = lat_a[:, 1 + e * A:1 + (e + 1) * A]                                                     # [G, A, D]
        want = torch.relu(zp[:, e, None, :] + table.t()[None])
        np.testing.assert_allclose(kids.numpy(), want.numpy(), rtol=1e-5, atol=2e-6)
    with pytest.raises(NotImplementedError):
        b.run(*args, fused=False)


@pytest.mark.parametrize("which,G,S", [("c4", 100, 50), ("c4", 4100, 5), ("ragged", 35, 12)])
@pytest.mark.parametrize("keep", [False, True])
def test_written_out_schedule_equals_the_general_one(which, G, S, keep):
    """Where every wavefront has a full share of every layer's column tiles the one-launch search runs a schedule
    written out for those shares (a third register slot for weight tiles, a merge tile kept from the leaf's merge
    to the drawn child's, further merge tiles resident in LDS); mz_search_seq_three_slots(0) makes the same
    searches take the general schedule.  Pools and results are identical bit for bit, with and without stored
    latents (src/search/mcts.py:44-74 either way)."""
    from muzero_amd import _hip
    from muzero_amd.engine import BatchedMCTS, PackedNetworks
    if which == "c4":
        _, g, f = _c4_nets(5)
        D = 168
    else:
        g, f = _custom_nets((1, 10, 17), 6, [128], [256, 128], [256, 256], ("relu",))
        D = 170
    A = g.num_actions
    nets = PackedNetworks(g, f)
    lat, eta, rand, to_play = _inputs(G, A, S, D, seed=7 * G + S)
    args = [torch.from_numpy(x).cuda() for x in (lat, to_play, eta, rand)]
    lib = _hip.lib()
    runs = []
    old = lib.mz_search_seq_three_slots(-1)
    assert old == 1
    try:
        for choice in (1, 0):
            lib.mz_search_seq_three_slots(choice)
            m = BatchedMCTS(nets, G, S, 1, discount=0.997, noise_frac=0.25, keep_latents=keep)
            res = m.run(*args)
            torch.cuda.synchronize()
            runs.append((m, res))
    finally:
        lib.mz_search_seq_three_slots(old)
    (a, ra), (b, rb) = runs
    names = ["visit", "value_sum", "prior", "reward", "parent", "first_child", "to_play", "n_nodes"] + (["latent"] if keep else [])
    for name in names:
        assert torch.equal(getattr(a.pool, name), getattr(b.pool, name)), name
    assert torch.equal(a.z_pool, b.z_pool)
    assert torch.equal(ra.policy, rb.policy) and torch.equal(ra.value, rb.value) and torch.equal(ra.visits, rb.visits)
    assert torch.equal(ra.policy32, rb.policy32)
