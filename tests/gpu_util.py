"""Helpers shared by the GPU parity tests."""
import numpy as np
import torch


def random_streams(G, A, S, depth, seed=0, value_scale=0.1, tie_fraction=0.1):
    """Synthetic recorded network outputs: normalised U(0,1) priors (a fraction of trees get
    exactly equal priors to exercise first-index tie-breaking), N(0, value_scale) rewards
    and values, uniform child choices, Dirichlet(0.3)-like eta."""
    rng = np.random.default_rng(seed)
    p = rng.random((G, S + 1, A), dtype=np.float32) + 1e-3
    p /= p.sum(-1, keepdims=True)
    ties = rng.random(G) < tie_fraction
    p[ties] = np.float32(1.0 / A)
    rew = (rng.standard_normal((G, S + 1, A)) * value_scale).astype(np.float32)
    sim = (rng.standard_normal((G, S, depth + 1)) * value_scale).astype(np.float32)
    big = rng.random(G) < 0.25
    sim[big] *= 40.0  # value-dominated trees
    rand = rng.integers(0, A, size=(G, S + 1)).astype(np.int32)
    eta = rng.gamma(0.3, size=(G, A)).astype(np.float32) + 1e-6
    eta /= eta.sum(-1, keepdims=True)
    to_play = rng.choice(np.array([1, -1, 0], np.int8), size=G)
    return dict(prior=p, reward=rew, sim=sim, rand=rand, eta=eta.astype(np.float32), to_play=to_play)


def dev(a, device="cuda"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def streams_to_device(st):
    """[G, ...]-major numpy streams -> step-major device tensors used by engine.run_recorded."""
    return dict(
        prior=dev(st["prior"].transpose(1, 0, 2)), reward=dev(st["reward"].transpose(1, 0, 2)),
        rand=dev(st["rand"].T), sim=dev(st["sim"].transpose(1, 0, 2)), eta=dev(st["eta"]),
        to_play=dev(st["to_play"]))


def explain_divergence(a, b, gi, A, S, depth, to_play, eta, rand, noise, discount, policy_of_child=None, tol=1e-4):
    """Two runs of the same search (streams dicts with prior f32[G,S+1,A], reward f32[G,S+1,A], rollout
    f32[G,S,depth+1], leaf i32[G,S]) whose network outputs differ by rounding: finds tree gi's first simulation
    where run `a` leaves run `b` and proves that the decision was a near tie -- either
      "select": the two runs descend differently; replaying both runs' streams up to that simulation through the
                CPU tree oracle, the two candidates' PUCT scores (src/search/selection.py:43-54) in run b lie closer
                together than twice the largest score discrepancy between the runs at that node, itself below
                `tol` of the scores' magnitude; or
      "rollout": the rollouts differ by more than `tol` of their scale although the same child was simulated: the
                greedy action (src/search/simulation.py:36) flipped, and the child's two largest policy entries
                (policy_of_child(gi, s) -> f32[A] from run a) lie within 1e-5 of the largest.
    Returns None when the runs agree on every descent and every rollout, else (kind, simulation)."""
    from oracle import pyref, tree as otree
    scale = max(1.0, float(np.abs(b["rollout"][gi]).max()))
    s_star = kind = None
    for s in range(S):
        if a["leaf"][gi, s] != b["leaf"][gi, s]:
            s_star, kind = s, "select"
            break
        if np.abs(a["rollout"][gi, s] - b["rollout"][gi, s]).max() > tol * scale:
            s_star, kind = s, "rollout"
            break
    if s_star is None:
        return None
    if kind == "rollout":
        assert policy_of_child is not None, f"tree {gi}: rollout of simulation {s_star} differs and no policy was given"
        pol = np.sort(np.asarray(policy_of_child(gi, s_star), np.float64))[::-1]
        assert pol[0] - pol[1] <= 1e-5 * pol[0], (f"tree {gi}: rollout of simulation {s_star} differs without a near-tie "
                                                 f"greedy action (top two policy entries {pol[0]}, {pol[1]})")
        return kind, s_star

    def before(run):  # the tree right before simulation s_star
        sl = slice(gi, gi + 1)
        return otree.run_batch(A, s_star, depth, to_play[sl], run["prior"][sl, :s_star + 1], run["reward"][sl, :s_star + 1],
                               rand[sl, :s_star + 1], run["rollout"][sl, :s_star], eta[sl], noise, discount)
    ta, tb = before(a), before(b)
    # the path run a takes: root .. its leaf (parents from a's own tree; the nodes exist in both trees: same expansions so far)
    path = [int(a["leaf"][gi, s_star])]
    while path[-1] != 0:
        path.append(int(ta["parent"][0, path[-1]]))
    path.reverse()

    def scores(t, node):
        fc = int(t["first_child"][0, node])
        sq = np.sqrt(float(t["visit"][0, node]))
        out = []
        for c in range(fc, fc + A):
            sh = pyref.PNode(None, None, 0, c)
            sh.visit, sh.value_sum = int(t["visit"][0, c]), float(t["value_sum"][0, c])
            sh.prior = np.float32(t["prior"][0, c]) if node == 0 else float(t["prior"][0, c])
            out.append(float(pyref._puct_score(sh, sq)))
        return np.array(out), fc
    node = 0
    for nxt in path[1:]:
        sb, fc = scores(tb, node)
        sa, _ = scores(ta, node)
        best, took = int(np.argmax(sb)), nxt - fc
        if took != best:
            disc = float(np.abs(sa - sb).max())
            gap = float(sb[best] - sb[took])
            mag = max(1.0, float(np.abs(sb).max()))
            assert disc <= tol * mag, f"tree {gi}: PUCT scores of the two runs differ by {disc}"
            assert gap <= 2 * disc + 1e-12, (f"tree {gi}, simulation {s_star}: the runs select differently by a score gap of "
                                             f"{gap}, discrepancy only {disc}")
            return kind, s_star
        node = nxt
    raise AssertionError(f"tree {gi}: leaves differ at simulation {s_star} although every level agrees")
