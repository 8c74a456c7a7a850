"""Lock-step search with residual-tower networks at the shipped latent shape
(src/config/config_car_racing.yaml:57-165: latent [256, 6, 6], one ResBlock 256 -> 3 in g and f,
reward [128], value [256, 128], policy [256, 256]; 5 actions, 50 simulations per move).

    python tools/bench_towers.py [--games 256] [--sims 50] [--steps 3] [--latent 256 6 6]

Prints one JSON line: simulations/s of the whole search, and for the dominant kernel (the
fc_merge product, fp32 MFMA) its achieved TFLOP/s measured with HIP events around the call.
Synthetic inputs, random-init weights (torch default init, seed 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

MFMA_F32_PEAK_TFLOPS = 157.3
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pmc_traffic():
    """HBM bytes per launch of the fc_merge kernel from the committed counter passes (None if absent)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r1_g_wide_pmc_traffic.json")) as fh:
            return json.load(fh)["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def build_nets(shape, A):
    from muzero_amd.config import DenseLayerConfig, DynamicsNetworkConfig, PredictionNetworkConfig, ResBlockConfig
    from muzero_amd.neural_networks import DynamicsNetwork, PredictionNetwork
    d = lambda n: DenseLayerConfig(out_features=n, activation="relu")  # noqa: E731
    rb = lambda: ResBlockConfig(out_channels=3, kernel_size=3, stride=1, padding=1, activation="relu",  # noqa: E731
                                pool_kernel_size=1, pool_stride=1)
    torch.manual_seed(0)
    g = DynamicsNetwork(shape, A, DynamicsNetworkConfig(res_net=[rb()], reward_net=[d(128)]))
    f = PredictionNetwork(shape, A, PredictionNetworkConfig(res_net=[rb()], value_net=[d(256), d(128)],
                                                            policy_net=[d(256), d(256)]))
    return g, f


BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 matrix peak of an MI355X


def merge_conv_roofline(nets, G, A, D, hw, C, seconds, flop):
    """Roofline entry of k_merge_conv's expansion call.  `flop`: the algorithmic float32 multiply-adds.  In the split
    form the kernel executes three fp16 products per product (operands as two fp16 planes; on 128 of the merge
    layer's reduction indices: pairs of k-tiles) on the fp16 matrix pipe: `achieved` / `peak` / `frac` then refer
    to that pipe, and `fp32_equivalent_tflops` says what the same result would need of the fp32 pipe (157.3 TFLOP/s
    peak)."""
    out = {"bound": "mfma", "kernel": "k_merge_conv [g expansion, search form]", "unit": "TFLOP/s",
           "avg_launch_us": seconds * 1e6, "flop_per_launch": flop, "fp32_equivalent_tflops": flop / seconds / 1e12,
           "fp32_mfma_peak_tflops": MFMA_F32_PEAK_TFLOPS}
    if getattr(nets, "split_merge_conv", False):
        kp32 = (nets.merge_t_perm.Kp + 31) // 32 * 32
        executed = 3.0 * (2.0 * G * kp32 * D + 2.0 * G * A * hw * C * 32)
        out.update({"kernel": "k_merge_conv_bf [g expansion, search form: operands as two fp16 planes, three products]",
                    "achieved": executed / seconds / 1e12, "peak": BF16_MFMA_PEAK_TFLOPS,
                    "frac": executed / seconds / 1e12 / BF16_MFMA_PEAK_TFLOPS, "executed_f16_flop_per_launch": executed,
                    "peak_note": "dense fp16 matrix peak (the same 2500 TFLOP/s as bf16); achieved = executed fp16 flops"})
    else:
        out.update({"achieved": flop / seconds / 1e12, "peak": MFMA_F32_PEAK_TFLOPS,
                    "frac": flop / seconds / 1e12 / MFMA_F32_PEAK_TFLOPS})
    return out


def cpu_network_time(g, f, shape, A, budget_s, threads):
    """The torch modules on the host, called as the reference calls them per simulation
    (src/search/expansion.py:25-30, src/search/simulation.py:31-44: 3 f calls at batch size 1, one g
    call at batch size 1 and one at batch size A).  Tree bookkeeping is not included (microseconds
    against the 18432 x 9216 fc_merge product of every g call)."""
    import copy
    g, f = copy.deepcopy(g).cpu(), copy.deepcopy(f).cpu()
    torch.set_num_threads(threads)
    lat1, latA = torch.randn(1, *shape), torch.randn(A, *shape)
    a1, aA = torch.zeros(1, dtype=torch.long), torch.arange(A)
    per_sim = []
    with torch.no_grad():
        f(lat1), g(lat1, a1)  # first-use costs
        t_end = time.perf_counter() + budget_s
        while time.perf_counter() < t_end or not per_sim:
            t0 = time.perf_counter()
            f(lat1)
            g(latA, aA)
            f(lat1)
            g(lat1, a1)
            f(lat1)
            per_sim.append(time.perf_counter() - t0)
    mean = sum(per_sim) / len(per_sim)
    return {"value": 1.0 / mean, "unit": "sims/s", "cores": threads, "kind": "port",
            "sample": f"network calls of {len(per_sim)} simulations (3 f at batch 1, g at batch 1 and {A}) on the "
                      f"torch modules, {threads} host threads, {sum(per_sim):.1f}s"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=256)
    ap.add_argument("--sims", type=int, default=50)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--actions", type=int, default=5)
    ap.add_argument("--latent", type=int, nargs=3, default=[256, 6, 6])
    ap.add_argument("--cpu-seconds", type=float, default=0.0, help="time the torch modules on the host for this long")
    ap.add_argument("--plain-root", action="store_true",
                    help="expand the roots from their 9216-wide latents (default: from the bottleneck values of the "
                         "representation network's final 1x1 convolution, as self-play with the shipped h does)")
    args = ap.parse_args()
    from muzero_amd.engine import BatchedMCTS
    from muzero_amd.towers import TowerNetworks
    shape, A, G, S = tuple(args.latent), args.actions, args.games, args.sims
    D = shape[0] * shape[1] * shape[2]
    t0 = time.time()
    g, f = build_nets(shape, A)
    cpu = None
    if args.cpu_seconds > 0:
        cpu = cpu_network_time(g, f, shape, A, args.cpu_seconds, min(16, os.cpu_count() or 1))
        t0 = time.time()
    # the shipped representation network ends in ResBlock(256 -> 3), a 1x1 convolution 3 -> 256 and an adaptive
    # average pool (src/config/config_car_racing.yaml:117-131, neural_network.py:51,60-61): a root latent is that
    # convolution of 3 x 6 x 6 pooled values, and the synthetic roots are drawn in that form
    torch.manual_seed(1)
    fin = torch.nn.Conv2d(3, shape[0], 1).cuda() if shape[0] != 3 else None
    nets = TowerNetworks(g, f, root_final=fin)
    torch.cuda.synchronize()
    t_build = time.time() - t0
    mcts = BatchedMCTS(nets, G, S, 1, discount=0.998, noise_frac=0.25, keep_latents=False)
    gen = torch.Generator(device="cuda").manual_seed(0)
    root_t = torch.randn(G, 3, shape[1], shape[2], device="cuda", generator=gen) if fin is not None else None
    with torch.no_grad():
        lat = fin(root_t).reshape(G, D).contiguous() if fin is not None else torch.randn(G, D, device="cuda", generator=gen)
    use_root_t = root_t is not None and not args.plain_root and mcts.root_form_ok(root_t)
    root_kw = {"root_t": root_t} if use_root_t else {}
    eta = torch.distributions.Dirichlet(torch.full((A,), 0.3)).sample((G,)).float().cuda()
    rand = torch.randint(0, A, (G, S + 1), device="cuda", generator=gen, dtype=torch.int32)
    to_play = torch.zeros(G, dtype=torch.int8, device="cuda")
    warm_s = []  # the first search runs launch by launch, the second records the graph
    for _ in range(args.warmup):
        torch.cuda.synchronize()
        t0 = time.time()
        res = mcts.run(lat, to_play, eta, rand, **root_kw)
        torch.cuda.synchronize()
        warm_s.append(time.time() - t0)
    t0 = time.time()
    for _ in range(args.steps):
        res = mcts.run(lat, to_play, eta, rand, **root_kw)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / args.steps
    assert int(res.visits.sum()) == G * S
    # the dominant kernel alone: fc_merge's latent half for G rows (one call per expansion and
    # one per rollout step, i.e. 2 per simulation)
    base = torch.empty(G, D, device="cuda")
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n_rep = 20
    nets._wide(nets.merge, lat, D, None, base, G)
    ev0.record()
    for _ in range(n_rep):
        nets._wide(nets.merge, lat, D, None, base, G)
    ev1.record()
    torch.cuda.synchronize()
    merge_ms = ev0.elapsed_time(ev1) / n_rep
    merge_flop = 2.0 * G * D * D
    # whole g / f stages
    def timed(fn):
        fn()
        ev0.record()
        for _ in range(5):
            fn()
        ev1.record()
        torch.cuda.synchronize()
        return ev0.elapsed_time(ev1) / 5
    rows = mcts.tree_base.clone()
    pol = torch.empty(G, A, device="cuda")
    val = torch.empty(G, device="cuda")
    arg = torch.empty(G, dtype=torch.int32, device="cuda")
    rew = torch.empty(G, A, device="cuda")
    nxt = torch.empty(G * A, D, device="cuda")
    f_ms = timed(lambda: nets.prediction(mcts.pool.latent, G, rows=rows, softmax_times=1, policy=pol))
    gA_ms = timed(lambda: nets.dynamics(mcts.pool.latent, G * A, rows=rows.repeat_interleave(A), action=None,
                                        next_latent=nxt, reward=rew))
    g1_ms = timed(lambda: nets.dynamics(mcts.pool.latent, G, rows=rows, action=arg.zero_(), next_latent=nxt[:G],
                                        reward=val))
    # the kernels that take most of a step of the search form, each alone: fc_merge on the bottleneck values fused
    # with the first product of g's tower (expansion: every leaf with all A actions; rollout step: one action per
    # tree), against the fp32 matrix peak, and the one-wave tower kernel's calls
    tower_roofline = None
    if getattr(nets, "fused_g", False) and nets.tower_f_t is not None:
        dt_ = nets.bottleneck_g
        hw = shape[1] * shape[2]
        t = torch.randn(G, dt_, device="cuda")
        t_out = torch.empty(G * A, dt_, device="cuda")
        act = torch.zeros(G, dtype=torch.int32, device="cuda")
        exp_ms = timed(lambda: nets.merge_conv(t, None, None, G))
        step_ms = timed(lambda: nets.merge_conv(t, None, act, G))
        exp_flop = 2.0 * G * nets.merge_t_perm.Kp * D + 2.0 * G * A * hw * shape[0] * 32
        step_flop = 2.0 * G * nets.merge_t_perm.Kp * D + 2.0 * G * hw * shape[0] * 32
        f_t = torch.empty(G, nets.bottleneck_f, device="cuda")
        thin_f_ms = timed(lambda: nets._tower_call(nets.tower_f_t, t, dt_, None, None, None, None, 0, None, G, 1, t_out=f_t))
        tower_roofline = merge_conv_roofline(nets, G, A, D, hw, shape[0], exp_ms * 1e-3, exp_flop)
        tower_roofline.update({"rollout_step_call": {"avg_launch_us": step_ms * 1e3, "flop_per_launch": step_flop,
                                                     "fp32_equivalent_tflops": step_flop / step_ms / 1e9},
                               "one_wave_tower_kernel_f_call_us": thin_f_ms * 1e3, "traffic": None})
    # reference arithmetic per simulation: 3 f calls and A+1 g calls, each g with the full 2D x D fc_merge
    F_heads = 2.0 * (D * 256 + 256 * 128 + 128 + D * 256 + 256 * 256 + 256 * A)
    F_tower = 2.0 * 36 * (shape[0] * 9 * 3 + 3 * 9 * 3 + shape[0] * 3 + 3 * shape[0])
    F_f = F_tower + F_heads
    F_g = 2.0 * 2 * D * D + F_tower + 2.0 * (D * 128 + 128)
    print(json.dumps({
        "metric": "mcts_simulations_per_sec", "value": G * S / dt, "unit": "sims/s", "n_gpus": 1,
        "ms_per_step": dt * 1e3, "steps": args.steps, "dtype": "f32+f64", "data": "synthetic",
        "config": {"workload": f"{G} trees x {S} simulations, lock-step search, residual-tower g/f, latent "
                               f"{list(shape)}, {A} actions",
                   "roots": "bottleneck values of h's final 1x1 convolution" if use_root_t else "full latents"},
        "reference_flop_per_sim": 3 * F_f + (A + 1) * F_g,
        "algorithmic_tflops": (3 * F_f + (A + 1) * F_g) * G * S / dt / 1e12,
        "stage_ms": {"f(G rows)": f_ms, "g expansion (G*A rows)": gA_ms, "g step (G rows)": g1_ms},
        "roofline": {"bound": "mfma", "kernel": "k_wide<4,4,true> + k_wide_finish [fc_merge latent half]",
                     "achieved": merge_flop / merge_ms / 1e9, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": merge_flop / merge_ms / 1e9 / MFMA_F32_PEAK_TFLOPS, "avg_launch_us": merge_ms * 1e3,
                     "flop_per_launch": merge_flop, "traffic": pmc_traffic() if (G, D) == (256, 9216) else None},
        "roofline_tower": tower_roofline,
        "build_s": t_build, "warmup_search_s": warm_s, "cpu_baseline": cpu,
    }))


if __name__ == "__main__":
    main()
