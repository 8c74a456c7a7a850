"""Tree kernels alone (select / expand-commit / backup in lock step), driven by recorded
network outputs: priors ~ normalised U(0,1), rewards and rollout values ~ N(0, 0.1), seed 0.

    python tools/bench_tree.py [--games 4096 65536 524288] [--sims 50]

Prints one JSON line per game count: simulations/s of the tree path and its achieved HBM rate
for the algorithmic bytes of one simulation,
    20 A d + 4 d   (select: A x (visit i32, value_sum f64, prior f64) and the child index per level)
  + 32 A           (child records of one expansion)
  + 24 (d + 2)     (visit / value_sum read-modify-write along the path)
with d = the measured mean depth of the selected leaf, against the 8 TB/s HBM peak.  The
per-kernel times come from HIP events around each kind of launch.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HBM_PEAK_GBPS = 8000.0


def run(G: int, S: int, A: int, depth: int = 1) -> dict:
    from muzero_amd.engine import TreePool
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(0)
    pool = TreePool(G, A, S, 1, with_latent=False)
    prior = torch.rand((S + 1, G, A), device=dev, generator=gen)
    prior = (prior / prior.sum(-1, keepdim=True)).contiguous()
    reward = (torch.randn((S + 1, G, A), device=dev, generator=gen) * 0.1).contiguous()
    rand = torch.randint(0, A, (S + 1, G), device=dev, generator=gen, dtype=torch.int32)
    sim = (torch.randn((S, G, depth + 1), device=dev, generator=gen) * 0.1).contiguous()
    eta = torch.distributions.Dirichlet(torch.full((A,), 0.3)).sample((G,)).float().to(dev)
    to_play = torch.ones(G, dtype=torch.int8, device=dev)
    leaf = torch.zeros(G, dtype=torch.int32, device=dev)
    child = torch.zeros(G, dtype=torch.int32, device=dev)
    depth_out = torch.zeros(G, dtype=torch.int32, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    t_sel = t_exp = t_bak = 0.0
    depth_sum = 0.0

    def once(timed: bool):
        nonlocal t_sel, t_exp, t_bak, depth_sum
        pool.init(to_play)
        pool.expand_commit(leaf.zero_(), prior[0], reward[0], rand[0], child)
        pool.root_noise(eta, 0.25)
        for s in range(S):
            ev[0].record()
            pool.select(leaf, depth_out)
            ev[1].record()
            pool.expand_commit(leaf, prior[s + 1], reward[s + 1], rand[s + 1], child)
            ev[2].record()
            pool.backup(child, leaf, sim[s], 0.998)
            ev[3].record()
            if timed:
                torch.cuda.synchronize()
                t_sel += ev[0].elapsed_time(ev[1])
                t_exp += ev[1].elapsed_time(ev[2])
                t_bak += ev[2].elapsed_time(ev[3])
                depth_sum += float(depth_out.float().mean())

    once(False)
    once(True)
    # untimed-inside run for the end-to-end rate
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    once(False)
    e1.record()
    torch.cuda.synchronize()
    total_ms = e0.elapsed_time(e1)
    d = depth_sum / S
    bytes_per_sim = 20 * A * d + 4 * d + 32 * A + 24 * (d + 2)
    sims = G * S

    # the same search with the tree side of a step as ONE launch (child records + backup + next descent)
    def once_fused():
        pool.init(to_play)
        pool.expand_commit(leaf.zero_(), prior[0], reward[0], rand[0], child)
        pool.root_noise(eta, 0.25)
        pool.select(leaf)
        for s in range(S):
            pool.commit_backup_select(leaf, prior[s + 1], reward[s + 1], rand[s + 1], child, sim[s], 0.998,
                                      select_next=s + 1 < S)

    visits_separate = pool.visit.clone()
    once_fused()
    torch.cuda.synchronize()
    assert torch.equal(pool.visit, visits_separate), "the one-launch step built other trees"
    e0.record()
    once_fused()
    e1.record()
    torch.cuda.synchronize()
    fused_ms = e0.elapsed_time(e1)
    return {"metric": "tree_simulations_per_sec", "value": sims / total_ms * 1e3, "unit": "sims/s", "games": G,
            "sims_per_move": S, "actions": A, "mean_leaf_depth": d, "algorithmic_bytes_per_sim": bytes_per_sim,
            "ms_per_search": total_ms,
            "one_launch_per_step": {"value": sims / fused_ms * 1e3, "ms_per_search": fused_ms,
                                    "us_per_launch": fused_ms / S * 1e3,
                                    "hbm_frac": sims * bytes_per_sim / fused_ms / 1e6 / HBM_PEAK_GBPS},
            "kernel_us_per_launch": {"k_select": t_sel / S * 1e3, "k_expand_commit": t_exp / S * 1e3,
                                     "k_backup": t_bak / S * 1e3},
            "roofline": {"bound": "hbm", "achieved": sims * bytes_per_sim / total_ms / 1e6, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": sims * bytes_per_sim / total_ms / 1e6 / HBM_PEAK_GBPS},
            "data": "synthetic recorded network outputs"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, nargs="+", default=[4096, 65536, 524288])
    ap.add_argument("--sims", type=int, default=50)
    ap.add_argument("--actions", type=int, default=7)
    args = ap.parse_args()
    for G in args.games:
        print(json.dumps(run(G, args.sims, args.actions)), flush=True)


if __name__ == "__main__":
    main()
