"""MCTS with the reference's constructor and `run(root)` contract (src/search/mcts.py:17-103),
executed by the batched GPU engine with a batch of one tree.

`run` consumes the global RNGs exactly like the reference -- one discarded `random.choice`
for the root expansion, one torch Dirichlet sample, one `random.choice` per simulation --
by drawing those numbers up front and handing them to the kernels as streams.  Afterwards the
whole tree is materialised under `root` as `Node` objects (children dicts in action order,
visit counts, value sums, priors, rewards, latents), which is what callers and the
reference's tests read.
"""
from __future__ import annotations

import math
import random
import time

import torch

from muzero_amd.search.nodes import Node


class MCTS:
    def __init__(self, selection, simulation, backpropagation, dynamic_network, prediction_network,
                 actions: torch.Tensor, max_itr: int = 0, max_time: float = 0.0, dirichlet_alpha: float = 0.3,
                 noise_frac: float = 0.25, visualize: bool = False) -> None:
        self.selection = selection
        self.simulation = simulation
        self.backpropagation = backpropagation
        self.actions = actions
        self.dynamics_network = dynamic_network
        self.prediction_network = prediction_network
        self.max_itr = max_itr
        self.max_time = max_time
        self.dirichlet_alpha = dirichlet_alpha
        self.noise_frac = noise_frac
        self.visualize = visualize
        self._engine = None

    # ------------------------------------------------------------------ engine plumbing
    def _get_engine(self, num_sims: int):
        from muzero_amd.engine import BatchedMCTS
        from muzero_amd.search.expansion import packed_for
        nets = packed_for(self.dynamics_network, self.prediction_network)
        depth = getattr(self.simulation, "depth", 1)
        discount = getattr(self.backpropagation, "discount", 1.0)
        key = (id(nets), num_sims, depth, discount, self.noise_frac)
        if self._engine is None or self._engine[0] != key:
            self._engine = (key, BatchedMCTS(nets, 1, num_sims, depth, discount, self.noise_frac, device=nets.device))
        return self._engine[1]

    def run(self, root: Node) -> tuple[list[float], float]:
        from muzero_amd.search.selection import PUCT
        A = int(self.actions.size(0))
        if A <= 0:
            raise ValueError("No possible actions to expand the node with")
        if not isinstance(self.selection, PUCT):
            return self._run_host_strategies(root)
        if self.max_itr == 0:
            return self._run_timed(root, A)
        eng = self._get_engine(self.max_itr)
        rand = [random.randrange(A)]                      # root expansion's discarded choice
        eta = torch.distributions.Dirichlet(torch.full((A,), self.dirichlet_alpha)).sample()
        rand += [random.randrange(A) for _ in range(self.max_itr)]
        dev = eng.pool.device
        res = eng.run(root.latent_state.detach().to(dev, torch.float32).reshape(1, -1),
                      torch.tensor([int(root.to_play)], dtype=torch.int8, device=dev), eta.reshape(1, A).to(dev),
                      torch.tensor([rand], dtype=torch.int32, device=dev))
        self._materialise(eng.pool, root)
        return res.policy[0].tolist(), float(res.value[0].item())

    def _run_timed(self, root: Node, A: int):
        """max_iterations == 0: search for `max_time` seconds (mcts.py:58-61).  Wall-clock
        bounded, hence not reproducible; runs in slices of simulations on the same kernels."""
        budget, done, slice_len = float(self.max_time), 0, 16
        cap_sims = 4096
        eng = self._get_engine(cap_sims)
        dev = eng.pool.device
        rand0 = torch.tensor([[random.randrange(A)]], dtype=torch.int32, device=dev)
        eta = torch.distributions.Dirichlet(torch.full((A,), self.dirichlet_alpha)).sample().reshape(1, A).to(dev)
        eng.begin(root.latent_state.detach().to(dev, torch.float32).reshape(1, -1),
                  torch.tensor([int(root.to_play)], dtype=torch.int8, device=dev), eta, rand0)
        t0 = time.time()
        while time.time() - t0 < budget and done + slice_len <= cap_sims:
            rand = torch.tensor([[random.randrange(A) for _ in range(slice_len)]], dtype=torch.int32, device=dev)
            eng.advance(rand)
            torch.cuda.synchronize()
            done += slice_len
        res = eng.finish()
        self._materialise(eng.pool, root)
        return res.policy[0].tolist(), float(res.value[0].item())

    def _run_host_strategies(self, root: Node):
        """Non-PUCT selection (UCT): the reference's loop with the strategy objects as given
        (mcts.py:50-74).  With UCT this raises `ValueError: math domain error` on the first
        step, as the reference does."""
        from muzero_amd.search.expansion import expand_node
        self._add_dirichlet_noise(root)
        itr = 0
        start = time.time()
        while (itr < self.max_itr) if self.max_itr else (time.time() - start < self.max_time):
            chosen = self.selection(root)
            expanded = expand_node(chosen, self.actions, self.dynamics_network, self.prediction_network)
            rewards = self.simulation(expanded)
            self.backpropagation(expanded, rewards, chosen.to_play)
            itr += 1
        utility = root.value_sum / root.visit_count
        return _soft_max([c.value_sum for c in root.children.values()]), utility

    def _add_dirichlet_noise(self, root: Node) -> None:
        from muzero_amd.search.expansion import expand_node
        expand_node(root, self.actions, self.dynamics_network, self.prediction_network)
        noise = torch.distributions.Dirichlet(torch.full((len(root.children),), self.dirichlet_alpha)).sample()
        for child, eta in zip(root.children.values(), noise):
            child.policy_priority += (1.0 - self.noise_frac) * child.policy_priority + self.noise_frac * eta

    # ------------------------------------------------------------------ pool -> Node objects
    def _materialise(self, pool, root: Node) -> None:
        n = int(pool.n_nodes[0].item())
        visit = pool.visit[0, :n].cpu().tolist()
        value = pool.value_sum[0, :n].cpu().tolist()
        prior = pool.prior[0, :n].cpu()
        reward = pool.reward[0, :n].cpu()
        first_child = pool.first_child[0, :n].cpu().tolist()
        to_play = pool.to_play[0, :n].cpu().tolist()
        latent = pool.latent[0, :n].to(root.latent_state.device)
        shape = (1,) + tuple(root.latent_state.shape[1:])
        A = pool.A
        nodes: list[Node] = [root] + [None] * (n - 1)
        root.visit_count, root.value_sum = visit[0], value[0]
        root.children = {}
        for i in range(n):
            fc = first_child[i]
            if fc < 0:
                continue
            parent = nodes[i]
            for a in range(A):
                c = fc + a
                child = Node(latent[c].reshape(shape), parent=parent, to_play=to_play[c])
                child.visit_count, child.value_sum = visit[c], value[c]
                # root children carry float32 tensors after the noise step, deeper ones floats
                child.policy_priority = prior[c].to(torch.float32) if i == 0 else float(prior[c])
                child.reward = reward[c:c + 1].clone()
                parent.children[self.actions[a]] = child
                nodes[c] = child


def _soft_max(values: list[float]) -> list[float]:
    ex = [math.exp(x) for x in values]
    total = sum(ex)
    return [x / total for x in ex]
