"""Host-side node object with the reference's fields (src/search/nodes.py:7-30).

Inside a search the tree lives in the device node pool; `Node` objects are what callers
hand in (the root) and read back (children, visit counts, value sums)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional, Tuple

import torch


@dataclass(eq=False)
class Node:
    latent_state: torch.Tensor
    parent: Optional["Node"] = None
    children: dict = field(default_factory=dict)
    to_play: int = 0
    visit_count: int = 0
    value_sum: float = 0.0
    reward: float = 0.0
    policy_priority: float = 0.0
    pos: Tuple[int, int] = (400, 100)

    def add_child(self, latent_state: torch.Tensor, action: torch.Tensor) -> "Node":
        child = Node(latent_state, parent=self, to_play=-self.to_play)
        self.children[action] = child
        return child
