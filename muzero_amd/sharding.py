"""Game sharding across ranks: one process per GPU, every rank owns a contiguous block of
games and searches them with no data-path collective (trees never interact).  The only
collectives are the barrier and the max-over-ranks of a timed region, so the same code runs
on RCCL (`nccl` backend, GPUs) and on `gloo` (CPU tests).

Reference: the self-play loop is one game per process (src/training_data_generator.py:93-146);
sharding whole games keeps every game's arithmetic identical to the single-process run.
"""
from __future__ import annotations

import os
import time
from dataclasses import dataclass

import torch


@dataclass(frozen=True)
class Shard:
    rank: int
    world: int
    local_rank: int
    games_per_rank: int

    @property
    def first_game(self) -> int:
        """Global index of this rank's first game: per-game random streams are seeded by the
        global index, so an N-rank job plays the games a 1-rank job of N*G games would."""
        return self.rank * self.games_per_rank

    @property
    def total_games(self) -> int:
        return self.world * self.games_per_rank

    def global_ids(self):
        return range(self.first_game, self.first_game + self.games_per_rank)


def shard_from_env(games_per_rank: int) -> Shard:
    """Reads RANK / WORLD_SIZE / LOCAL_RANK as torch.distributed.run exports them."""
    return Shard(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
                 int(os.environ.get("LOCAL_RANK", "0")), games_per_rank)


def fence(dist=None, cuda: bool = True):
    """Device sync + barrier + device sync: brackets a timed region on every rank."""
    if cuda:
        torch.cuda.synchronize()
    if dist is not None and dist.is_initialized():
        dist.barrier()
        if cuda:
            torch.cuda.synchronize()


def timed_steps(step, steps: int, dist=None, cuda: bool = True, device=None) -> float:
    """Runs `step` `steps` times between two fences and returns the MAX elapsed seconds over all
    ranks (the job is as slow as its slowest rank)."""
    fence(dist, cuda)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence(dist, cuda)
    dt = time.perf_counter() - t0
    return max_over_ranks(dt, dist, device)


def max_over_ranks(value: float, dist=None, device=None) -> float:
    if dist is None or not dist.is_initialized():
        return float(value)
    on_cpu = device is None or dist.get_backend() == "gloo"
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if on_cpu else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def whole_job_rate(units_per_rank_step: int, steps: int, world: int, seconds: float) -> float:
    """Aggregate throughput of the job: what all ranks processed / the slowest rank's time."""
    return world * units_per_rank_step * steps / seconds
