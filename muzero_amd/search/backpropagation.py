"""Backup strategy (src/search/backpropagation.py:5-18) on the `k_backup` kernel."""
from __future__ import annotations

import torch

from muzero_amd.search.nodes import Node


class Backpropagation:
    def __init__(self, discount: float) -> None:
        self.discount = discount

    def __call__(self, leaf_node: Node, rewards: list[float], to_play: int) -> None:
        from muzero_amd.search import _bridge
        root = leaf_node
        while root.parent is not None:
            root = root.parent
        pool, order, index = _bridge.upload(root)
        dev = pool.device
        start = index[id(leaf_node)]
        # the kernel takes the sign reference from a node's to_play: park the requested value
        # in the spare slot behind the tree
        spare = len(order)
        pool.set_link((0, spare), to_play=int(to_play))
        child = torch.tensor([start], dtype=torch.int32, device=dev)
        ref = torch.tensor([spare], dtype=torch.int32, device=dev)
        rollout = torch.tensor([[float(r) for r in rewards]], dtype=torch.float32, device=dev)
        pool.backup(child, ref, rollout, self.discount)
        _bridge.download_stats(pool, order)
