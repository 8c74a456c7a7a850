"""Self-play moves per second with the shipped network shapes and host-side environments
(src/config/config_car_racing.yaml:57-165: h = conv 2x2/2 -> 2 ResBlocks(128) -> conv 2x2/2 ->
3 ResBlocks(256) -> max-pool -> ResBlock(3) on a 32-frame history of 3x96x96 observations; g / f as in
tools/bench_towers.py), 256 games per GPU, 50 simulations per move.

The simulators are stand-ins that return random frames (the Box2D / pygame packages of the reference's
adapters are not part of this build): what is timed is everything else of a move -- observation upload,
history tensor, h, the search, the action pick and the host step calls.

    python tools/bench_selfplay_towers.py [--games 256] [--moves 4] [--history 32]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class RandomFrames:
    """Duck-typed single-player environment producing uniform noise frames."""

    def __init__(self, gid: int, shape, actions: int, length: int = 1000):
        self.gen = torch.Generator().manual_seed(gid)
        self.shape, self.actions, self.length, self.t = shape, actions, length, 0

    def _obs(self):
        return torch.rand((1,) + self.shape, generator=self.gen)

    def get_to_play(self):
        return 0

    def get_action_space(self):
        return (self.actions,)

    def get_observation_space(self):
        return self.shape

    def get_state(self):
        return self._obs()

    def step(self, action):
        self.t += 1
        return self._obs(), 0.1 * action, self.t >= self.length

    def reset(self):
        self.t = 0
        return self._obs()


def shipped_networks(history: int):
    from muzero_amd.config import ConvLayerConfig, PoolLayerConfig, RepresentationNetworkConfig, ResBlockConfig
    from muzero_amd.neural_networks import RepresentationNetwork
    from tools.bench_towers import build_nets
    shape, A = (256, 6, 6), 5
    rb = lambda c: ResBlockConfig(out_channels=c, kernel_size=3, stride=1, padding=1, activation="relu",  # noqa: E731
                                  pool_kernel_size=1, pool_stride=1)
    conv = lambda c: ConvLayerConfig(out_channels=c, kernel_size=2, stride=2, padding=0, activation="relu")  # noqa: E731
    torch.manual_seed(0)
    h = RepresentationNetwork((3, 96, 96), shape, RepresentationNetworkConfig(
        history_length=history,
        downsample=[conv(128), rb(128), rb(128), conv(256), rb(256), rb(256), rb(256),
                    PoolLayerConfig(kernel_size=2, stride=2, pool_type="max")],
        res_net=[rb(3)]))
    g, f = build_nets(shape, A)
    return h, g, f, shape, A


def run(games: int = 256, moves: int = 4, history: int = 32, sims: int = 50, workers: int = 16, parts: bool = True):
    """Serial engine and two-half pipelined engine on the same stand-in simulators: env steps per second each."""
    from muzero_amd.config import MCTSConfig, TrainingDataGeneratorConfig
    from muzero_amd.environments.host_pool import HostEnvPool
    from muzero_amd.selfplay import PipelinedSelfPlay, SelfPlayEngine
    G = games
    h, g, f, shape, A = shipped_networks(history)
    cfg = TrainingDataGeneratorConfig(
        num_episodes=1, max_steps_per_episode=1000, total_time=0, epsilon=0.25, epsilon_decay=1.0, min_epsilon=0.25,
        mcts=MCTSConfig(selection_strategy="puct", max_iterations=sims, max_time=0, model_look_ahead=1,
                        discount_factor=0.998, dirichlet_alpha=0.3, noise_frac=0.25))
    eng = SelfPlayEngine(h, g, f, G, cfg, seed=0, device_rng=True,
                         env=HostEnvPool([RandomFrames(i, (3, 96, 96), A) for i in range(G)], workers=workers))
    for _ in range(2):
        eng.move()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(moves):
        eng.move()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / moves
    out = {
        "metric": "selfplay_env_steps_per_sec", "value": G / dt, "unit": "env steps/s", "n_gpus": 1,
        "ms_per_move": dt * 1e3, "moves": moves, "data": "synthetic (random-frame host simulators)",
        "config": {"workload": f"{G} games, {sims} simulations per move, history {history} x 3x96x96, "
                               f"shipped h/g/f shapes (latent {list(shape)}), {A} actions"},
        "simulations_per_sec": G * sims / dt}
    if parts:  # parts of a move, timed alone
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        obs, to_play = eng.env.observe()
        hist = eng.history.push(obs, eng.step_in_episode, None)
        eng.h(hist)
        ev0.record()
        for _ in range(3):
            eng.h(hist)
        ev1.record()
        torch.cuda.synchronize()
        h_ms = ev0.elapsed_time(ev1) / 3
        t1 = time.time()
        for _ in range(3):
            eng.env.step(eng.action)
            eng.env.observe()
        torch.cuda.synchronize()
        env_ms = (time.time() - t1) / 3 * 1e3
        out["parts_ms"] = {"h (HIP convolution path, per-game batch statistics)": h_ms,
                           "host step + observe + upload": env_ms,
                           "search and the rest": dt * 1e3 - h_ms - env_ms}
    eng.env.close()
    del eng
    torch.cuda.empty_cache()
    # the same games in two halves: the host steps one half while the GPU searches the other
    piped = PipelinedSelfPlay(h, g, f, [RandomFrames(i, (3, 96, 96), A) for i in range(G)], cfg, seed=0, device_rng=True,
                              workers=workers)
    for _ in range(2):
        piped.move()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(moves):
        piped.move()
    torch.cuda.synchronize()
    dtp = (time.time() - t0) / moves
    piped.close()
    out["pipelined"] = {"value": G / dtp, "unit": "env steps/s", "ms_per_move": dtp * 1e3, "speedup_over_serial": dt / dtp,
                        "form": "two halves of the games alternate between the host thread pool and the GPU (one stream, "
                                "one event per half)"}
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=256)
    ap.add_argument("--moves", type=int, default=4)
    ap.add_argument("--history", type=int, default=32)
    ap.add_argument("--sims", type=int, default=50)
    ap.add_argument("--workers", type=int, default=16)
    args = ap.parse_args()
    print(json.dumps(run(args.games, args.moves, args.history, args.sims, args.workers)))


if __name__ == "__main__":
    main()
