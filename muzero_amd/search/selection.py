"""Selection strategies (src/search/selection.py:8-59).

PUCT runs the `k_select` kernel.  UCT is kept on the host exactly as the reference computes
it: it cannot drive a search (log(0) on the first step raises `ValueError: math domain
error` in the reference too), so it has no kernel."""
from __future__ import annotations

from math import log, sqrt

import torch

from muzero_amd.search.nodes import Node


class UCT:
    def __init__(self, c: float = 1.41):
        self.c = c

    def uct_score(self, node: Node) -> float:
        exploit = node.reward / node.visit_count
        explore = self.c * sqrt(log(node.parent.visit_count)) / (node.visit_count + 1)
        return exploit + explore

    def __call__(self, root: Node) -> Node:
        while len(root.children) > 0:
            root = max(root.children.values(), key=self.uct_score)
        return root


class PUCT:
    """c1 = 1.25 and c2 = 19652 are what the search factory always uses
    (src/search/factory.py:37); the kernel's lookup tables are built for them."""

    def __init__(self, c1: float = 1.25, c2: float = 19652, discount: float = 0.99):
        if (c1, c2) != (1.25, 19652):
            raise NotImplementedError("the PUCT kernel tables are built for c1=1.25, c2=19652")
        self.c1, self.c2, self.discount = c1, c2, discount

    def __call__(self, root: Node) -> Node:
        if len(root.children) == 0:
            return root
        from muzero_amd.search import _bridge
        pool, order, _ = _bridge.upload(root)
        # root children hold float32 tensors only after MCTS.run perturbed them (mcts.py:94)
        root_f32 = any(isinstance(c.policy_priority, torch.Tensor) for c in root.children.values())
        leaf = torch.zeros(1, dtype=torch.int32, device=pool.device)
        pool.select(leaf, root_f32=root_f32)
        idx = int(leaf.item())
        if idx < 0:
            raise RuntimeError("selection failed: a visit count is outside the PUCT tables")
        return order[idx]
