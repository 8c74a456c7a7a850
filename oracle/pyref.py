"""TEST INFRASTRUCTURE ONLY -- per-node CPU port of the reference's MCTS + self-play step.

This is the timed CPU baseline ("port") of bench.py and a checker for tests; the product
package never imports it.  It keeps the reference's cost structure: Python node objects,
a Python max() over children per tree level, and the same five torch calls per
simulation at the same batch sizes (3x f at B=1, 1x g at B=A, 1x g at B=1 for
model_look_ahead=1).

Parity status: PINNED.  tests/test_oracle_golden.py checks it against streams and
per-node statistics captured from the imported reference (tools/make_goldens.py).

Reference call sites restated here:
  MCTS.run / _step / _add_dirichlet_noise / _soft_max   src/search/mcts.py:44-103
  PUCT.puct_score / __call__                            src/search/selection.py:43-59
  expand_node                                           src/search/expansion.py:9-35
  MuZeroSimulation.__call__                             src/search/simulation.py:23-45
  Backpropagation.__call__                              src/search/backpropagation.py:9-18
"""
from __future__ import annotations

import math
import random

import numpy as np
import torch

C1 = 1.25
C2 = 19652


class PNode:
    __slots__ = ("latent", "parent", "children", "to_play", "visit", "value_sum", "reward", "prior", "index")

    def __init__(self, latent, parent, to_play, index):
        self.latent = latent
        self.parent = parent
        self.children = []
        self.to_play = to_play
        self.visit = 0
        self.value_sum = 0.0
        self.reward = 0.0
        self.prior = 0.0  # python float below the root; numpy float32 for root children
        self.index = index


class TorchEvaluator:
    """Evaluates g and f exactly where the reference does (expansion.py:25-33,
    simulation.py:34-44)."""

    def __init__(self, dynamics, prediction, actions: torch.Tensor, depth: int):
        self.g = dynamics
        self.f = prediction
        self.actions = actions
        self.depth = depth

    def expand(self, latent):
        a = self.actions.size(0)
        policy, _ = self.f(latent)
        prior = torch.softmax(policy, dim=1).squeeze(0)
        nxt, rew = self.g(latent.repeat(a, 1, 1, 1), self.actions)
        return [prior[i].item() for i in range(a)], nxt, [rew[i].item() for i in range(a)]

    def rollout(self, latent):
        out = []
        s = latent
        for _ in range(self.depth):
            policy, _ = self.f(s)
            act = torch.tensor([policy.argmax().item()])
            s, r = self.g(s, act)
            out.append(r)
        _, v = self.f(s)
        out.append(v)
        return [x.item() for x in out]


class StreamEvaluator:
    """Replays recorded network outputs: prior[e][A], reward[e][A] per expansion e and
    the rollout list per simulation."""

    def __init__(self, prior_stream, reward_stream, sim_stream):
        self.prior = prior_stream
        self.reward = reward_stream
        self.sim = sim_stream
        self.e = 0
        self.s = 0

    def expand(self, latent):
        p = [float(x) for x in self.prior[self.e]]
        r = [float(x) for x in self.reward[self.e]]
        self.e += 1
        return p, [None] * len(p), r

    def rollout(self, latent):
        out = [float(x) for x in self.sim[self.s]]
        self.s += 1
        return out


class RecordingEvaluator:
    """Wraps an evaluator and keeps what it returned (to build streams)."""

    def __init__(self, inner):
        self.inner = inner
        self.prior, self.reward, self.sim = [], [], []

    def expand(self, latent):
        p, n, r = self.inner.expand(latent)
        self.prior.append(list(p))
        self.reward.append(list(r))
        return p, n, r

    def rollout(self, latent):
        out = self.inner.rollout(latent)
        self.sim.append(list(out))
        return out


def _puct_score(child: PNode, sqrt_np: float):
    n = child.visit
    pb_c = sqrt_np / (n + 1)
    pb_c *= math.log((n + C2 + 1) / C2) + C1
    if isinstance(child.prior, np.float32):
        # root children: the reference holds a 0-dim float32 tensor here, so the product
        # and the sum are float32 operations on float32-rounded operands
        ps = np.float32(pb_c) * child.prior
        return (np.float32(child.value_sum / n) + ps) if n > 0 else ps
    ps = pb_c * child.prior
    return (child.value_sum / n + ps) if n > 0 else (0 + ps)


def select(root: PNode) -> PNode:
    node = root
    while node.children:
        s = math.sqrt(node.visit)
        node = max(node.children, key=lambda c: _puct_score(c, s))
    return node


def expand(node: PNode, ev, nodes: list, rand_index) -> PNode:
    prior, nxt, rew = ev.expand(node.latent)
    for a in range(len(prior)):
        lat = None if nxt[a] is None else nxt[a].unsqueeze(0)
        ch = PNode(lat, node, -node.to_play, len(nodes))
        ch.prior = prior[a]
        ch.reward = rew[a]
        node.children.append(ch)
        nodes.append(ch)
    return node.children[rand_index(len(prior))]


def backup(child: PNode, rewards, leaf_to_play, discount: float) -> float:
    g = 0.0
    for r in reversed(rewards):
        g = r + discount * g
    node = child
    while node is not None:
        node.visit += 1
        node.value_sum += g if node.to_play == leaf_to_play else -g
        node = node.parent
    return g


def add_root_noise(root: PNode, eta, noise_frac: float) -> None:
    e32 = np.float32(noise_frac)
    for ch, n in zip(root.children, eta):
        p = ch.prior
        keep = np.float32((1.0 - noise_frac) * p)
        ch.prior = np.float32(np.float32(p) + np.float32(keep + np.float32(e32 * np.float32(n))))


def run_mcts(root_latent, to_play, ev, num_actions, num_sims, *, discount, noise_frac,
             dirichlet_alpha=0.3, eta=None, rand_index=None, trace=None):
    """One MCTS.run.  Returns (tree_policy list[float], utility float, nodes list[PNode]).

    `eta`: fixed Dirichlet sample (float32 sequence) or None to draw one from torch's
    global RNG as the reference does.  `rand_index(n)`: the child-choice stream; default
    random.randrange, which consumes Python's global RNG exactly like random.choice."""
    rand_index = rand_index or random.randrange
    root = PNode(root_latent, None, to_play, 0)
    nodes = [root]
    expand(root, ev, nodes, rand_index)
    if eta is None:
        eta = torch.distributions.Dirichlet(torch.full((num_actions,), dirichlet_alpha)).sample().numpy()
    add_root_noise(root, eta, noise_frac)
    for _ in range(num_sims):
        leaf = select(root)
        child = expand(leaf, ev, nodes, rand_index)
        rewards = ev.rollout(child.latent)
        g = backup(child, rewards, leaf.to_play, discount)
        if trace is not None:
            trace.append((leaf.index, child.index, g, list(rewards)))
    utility = root.value_sum / root.visit
    ex = [math.exp(c.value_sum) for c in root.children]
    tot = sum(ex)
    return [x / tot for x in ex], utility, nodes


def pool_arrays(nodes, num_actions):
    """Flatten a node list (creation order) into the SoA arrays the HIP pool uses."""
    n = len(nodes)
    out = {
        "visit": np.array([x.visit for x in nodes], dtype=np.int32),
        "value_sum": np.array([x.value_sum for x in nodes], dtype=np.float64),
        "prior": np.array([float(x.prior) for x in nodes], dtype=np.float64),
        "reward": np.array([float(x.reward) for x in nodes], dtype=np.float32),
        "parent": np.array([-1 if x.parent is None else x.parent.index for x in nodes], dtype=np.int32),
        "first_child": np.array([x.children[0].index if x.children else -1 for x in nodes], dtype=np.int32),
        "to_play": np.array([x.to_play for x in nodes], dtype=np.int8),
    }
    assert all(len(v) == n for v in out.values())
    return out
