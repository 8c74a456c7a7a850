"""Moves small host-side `Node` trees into a device node pool and back, so that the
strategy classes (PUCT, Backpropagation) run the same kernels on hand-built trees that
MCTS.run uses on its own pool."""
from __future__ import annotations

import numpy as np
import torch

from muzero_amd.engine import TreePool
from muzero_amd.search.nodes import Node


def flatten(root: Node):
    """Breadth-first numbering in which the children of every node are consecutive.
    All expanded nodes must have the same number of children (that number becomes A)."""
    order = [root]
    first_child = {}
    i = 0
    width = None
    while i < len(order):
        node = order[i]
        kids = list(node.children.values())
        if kids:
            if width is None:
                width = len(kids)
            elif len(kids) != width:
                raise ValueError("device tree kernels need the same child count at every expanded node")
            first_child[id(node)] = len(order)
            order.extend(kids)
        i += 1
    return order, first_child, width or 1


def upload(root: Node, device="cuda"):
    order, first_child, A = flatten(root)
    n = len(order)
    index = {id(nd): i for i, nd in enumerate(order)}
    max_visit = max(int(nd.visit_count) for nd in order)
    pool = TreePool(1, A, 0, D=1, device=device, with_latent=False, cap=max(n + A, 1 + A))
    if max_visit + 2 > pool.tab_len:
        pool.set_table_len(max_visit + 2)
    dev = pool.device
    pool.visit[0, :n] = torch.tensor([int(nd.visit_count) for nd in order], dtype=torch.int32, device=dev)
    pool.value_sum[0, :n] = torch.tensor([float(nd.value_sum) for nd in order], dtype=torch.float64, device=dev)
    pool.prior[0, :n] = torch.tensor([float(nd.policy_priority) for nd in order], dtype=torch.float64, device=dev)
    pool.first_child[0, :n] = torch.tensor([first_child.get(id(nd), -1) for nd in order], dtype=torch.int32, device=dev)
    pool.set_link((0, slice(0, n)),
                  parent=[-1 if nd.parent is None or id(nd.parent) not in index else index[id(nd.parent)] for nd in order],
                  to_play=[int(nd.to_play) for nd in order])
    pool.n_nodes[0] = n
    return pool, order, index


def download_stats(pool: TreePool, order) -> None:
    n = len(order)
    visit = pool.visit[0, :n].cpu().numpy()
    value = pool.value_sum[0, :n].cpu().numpy()
    for i, nd in enumerate(order):
        nd.visit_count = int(visit[i])
        # keep ints as ints where the caller stored ints and nothing changed (known-answer tests compare ==)
        nd.value_sum = float(value[i])
