"""N > 1 path on the CPU: two ranks over gloo shard the games, search their blocks with the
oracle (test infrastructure: the CPU restatement of src/search/mcts.py:44-74) and must reproduce,
game for game, what one rank computes for all games; the timing helper must report the
slowest rank's time on every rank."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

A, S, DEPTH, GAMES_PER_RANK = 5, 12, 1, 6


def _streams(global_ids):
    """Per-game input streams seeded by the GLOBAL game index."""
    out = {k: [] for k in ("prior", "reward", "rand", "sim", "eta", "to_play")}
    for g in global_ids:
        r = np.random.default_rng(1000 + g)
        p = r.random((S + 1, A)).astype(np.float32)
        out["prior"].append(p / p.sum(-1, keepdims=True))
        out["reward"].append(r.normal(size=(S + 1, A)).astype(np.float32))
        out["rand"].append(r.integers(0, A, size=S + 1).astype(np.int32))
        out["sim"].append(r.normal(size=(S, DEPTH + 1)).astype(np.float32))
        out["eta"].append(r.dirichlet(np.full(A, 0.3)).astype(np.float32))
        out["to_play"].append(np.int8(1 if g % 2 == 0 else -1))
    return {k: np.stack(v) for k, v in out.items()}


def _search(global_ids):
    from oracle import tree
    s = _streams(global_ids)
    return tree.run_batch(A, S, DEPTH, s["to_play"], s["prior"], s["reward"], s["rand"], s["sim"], s["eta"], 0.25, 0.997)


def _worker(rank, world, port, tmp):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from muzero_amd import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shard = sharding.shard_from_env(GAMES_PER_RANK)
        assert (shard.rank, shard.world, shard.total_games) == (rank, world, world * GAMES_PER_RANK)
        ids = list(shard.global_ids())
        # the blocks are disjoint and cover every game
        gathered = [None] * world
        dist.all_gather_object(gathered, ids)
        assert sorted(sum(gathered, [])) == list(range(shard.total_games))
        calls = []
        res = {}

        def step():
            calls.append(1)
            res.update(_search(ids))

        dt = sharding.timed_steps(step, 2, dist, cuda=False)
        assert len(calls) == 2 and dt > 0
        # the slowest rank's time on every rank
        assert sharding.max_over_ranks(float(rank + 1), dist) == float(world)
        assert sharding.whole_job_rate(GAMES_PER_RANK * S, 2, world, 2.0) == world * GAMES_PER_RANK * S
        np.savez(os.path.join(tmp, f"rank{rank}.npz"), visit=res["visit"], value_sum=res["value_sum"],
                 policy=res["policy"], utility=res["utility"], leaf=res["leaf"])
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_reproduce_the_single_rank_job(tmp_path):
    import oracle
    oracle.build()
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    whole = _search(range(world * GAMES_PER_RANK))
    for rank in range(world):
        part = np.load(tmp_path / f"rank{rank}.npz")
        sl = slice(rank * GAMES_PER_RANK, (rank + 1) * GAMES_PER_RANK)
        for k in ("visit", "value_sum", "policy", "utility", "leaf"):
            assert np.array_equal(part[k], whole[k][sl]), (rank, k)


def test_single_rank_helpers_need_no_process_group():
    from muzero_amd import sharding
    os.environ.pop("RANK", None), os.environ.pop("WORLD_SIZE", None), os.environ.pop("LOCAL_RANK", None)
    sh = sharding.shard_from_env(4096)
    assert (sh.rank, sh.world, sh.first_game, sh.total_games) == (0, 1, 0, 4096)
    assert sharding.max_over_ranks(1.5) == 1.5
    n = []
    assert sharding.timed_steps(lambda: n.append(1), 3, None, cuda=False) > 0 and len(n) == 3
