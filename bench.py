"""Headline benchmark: MCTS simulations/sec at 4096 concurrent Connect-Four games per GPU,
50 simulations per move (MLP h/g/f, A=7, D=168, model_look_ahead=1).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one MCTS.run over all G trees of this rank (root expansion + noise + 50
select/expand/simulate/backup rounds), inputs resident in HBM.  Games shard per GPU with no
data-path collective (weak scaling); torch.distributed is used for the barrier and the
max-over-ranks of the timed region only.  Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

A, D = 7, 168
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 matrix peak
F16_MFMA_PEAK_TFLOPS = 2500.0   # dense fp16 matrix peak (v_mfma_f32_16x16x32_f16: 16384 flop per 16 cycles on 1024 SIMDs at 2.4 GHz)
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X dense fp32 matrix peak
F16_PRODUCTS_PER_FP32_PRODUCT = 3  # operands as two fp16 planes: hi*hi, hi*lo, lo*hi
ROOFLINE_SCHEMA = 3  # 1: achieved = algorithmic fp32 flops / fp32 peak; 2: achieved = executed fp32 matrix flops / fp32 peak;
#                      3: achieved = algorithmic flops against the fp32-accurate rate of the fp16 pipe (fp16 peak / 3), executed beside it
HBM_PEAK_GBS = 8000.0


def build_networks():
    from muzero_amd.config import load_config
    from muzero_amd.neural_networks import DynamicsNetwork, PredictionNetwork, RepresentationNetwork
    cfg = load_config("config_c4_mlp.yaml")
    torch.manual_seed(0)
    h = RepresentationNetwork((2, 6, 7), cfg.networks.latent_shape, cfg.networks.representation)
    g = DynamicsNetwork(cfg.networks.latent_shape, A, cfg.networks.dynamics)
    f = PredictionNetwork(cfg.networks.latent_shape, A, cfg.networks.prediction)
    return cfg, h, g, f


def synthetic_roots(h, G, seed, device):
    """Root observations: Connect-Four positions after 0..15 random moves from the empty board
    (played by the GPU env kernels; games that ended early fall back to the empty board),
    assembled into the K=1 history tensor and passed through h."""
    from muzero_amd.environments.connect_four import BatchedConnectFour
    rng = np.random.default_rng(seed)
    env = BatchedConnectFour(G, device=device)
    env.reset()
    plies = torch.from_numpy(rng.integers(0, 16, size=G)).to(device)
    obs_t, tp_t, done_t = [], [], []
    for t in range(16):
        obs, to_play = env.observe()
        obs_t.append(obs.clone()), tp_t.append(to_play.clone()), done_t.append(env.done.clone())
        env.step(torch.from_numpy(rng.integers(0, 7, size=G).astype(np.int32)).to(device))
    obs_t, tp_t, done_t = torch.stack(obs_t), torch.stack(tp_t), torch.stack(done_t)
    idx = torch.arange(G, device=device)
    pick = torch.where(done_t[plies, idx].bool(), torch.zeros_like(plies), plies)
    obs, to_play = obs_t[pick, idx], tp_t[pick, idx].contiguous()
    hist = torch.cat([obs, torch.zeros(G, 1, 6, 7, device=device)], dim=1)  # frame + action plane (action id 0)
    with torch.no_grad():
        latent = h.to(device)(hist).reshape(G, D).contiguous()
    return latent, to_play


def child_choice_streams(G, S, base_seed):
    out = np.empty((G, S + 1), np.int32)
    for g in range(G):
        r = random.Random(base_seed + g)
        out[g] = [r.randrange(A) for _ in range(S + 1)]
    return out


def time_kernel(fn, iters=20):
    """Average duration (s) of `fn`'s launches on the current stream, by HIP events."""
    stream = torch.cuda.current_stream()
    fn()
    stream.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(iters):
        fn()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def mlp_flop(p):
    return sum(2 * p.layer[i].K * p.layer[i].N for i in range(p.n_layers))


def search_flop_per_tree(plan, S, depth):
    """Multiply-add work the fused search kernel executes per tree (2 flop per MAC): per
    expansion the policy head, the D x D merge product and A reward heads; per simulation
    `depth` greedy steps (policy head, merge product, reward head) and one value head."""
    expand = mlp_flop(plan.policy) + 2 * D * D + A * mlp_flop(plan.reward)
    simulate = depth * (mlp_flop(plan.policy) + 2 * D * D + mlp_flop(plan.reward)) + mlp_flop(plan.value)
    return (S + 1) * expand + S * simulate


def mlp_f16_flop(p):
    """fp16 matrix-instruction flops one row of an MLP costs: three products per padded (Kp x Np) weight block;
    one-output tail layers folded into the previous layer's epilogue run on the vector ALUs and count nothing."""
    return sum(F16_PRODUCTS_PER_FP32_PRODUCT * 2 * p.layer[i].Kp * p.layer[i].Np for i in range(p.n_layers) if p.layer[i].fmt == 1)


def search_f16_flop_per_tree(plan, S, depth):
    """What the matrix pipe executes per tree in the two-plane fp16 form (rows are whole 16-row tiles at the
    benchmarked sizes: 16 trees per workgroup, 16 * A child rows)."""
    merge = mlp_f16_flop(plan.merge)
    expand = mlp_f16_flop(plan.policy) + merge + A * mlp_f16_flop(plan.reward)
    simulate = depth * (mlp_f16_flop(plan.policy) + merge + mlp_f16_flop(plan.reward)) + mlp_f16_flop(plan.value)
    return (S + 1) * expand + S * simulate


def reference_flop_per_sim(plan, depth):
    """Network flops per simulation in the reference's own formulation (what a per-node
    implementation evaluates): (2 + depth) full f calls -- both heads each time -- and
    A + depth full g calls with the merge layer over [latent ; embedding]
    (src/search/expansion.py:24-33, src/search/simulation.py:31-44).  For c4_mlp, depth 1:
    3 * 372,480 + 8 * 156,160 = 2,366,720."""
    f_call = mlp_flop(plan.policy) + mlp_flop(plan.value)
    g_call = 2 * (2 * D) * D + mlp_flop(plan.reward)
    return (2 + depth) * f_call + (A + depth) * g_call


TREE_BYTES_PER_SIM = 168 * 2 + 272          # select + child records + backup at A = 7, leaf depth 2
LATENT_BYTES_PER_SIM = 4 * D * (A + 2)      # stored latents: read the leaf, write A children, read the drawn child
IMPLICIT_LATENT_BYTES_PER_SIM = 4 * D * 2   # one merge row per expansion: read the parent's row, write the leaf's


def kernel_rooflines(mcts, plan, step, stored, stored_step):
    """Times (HIP events on the launch stream) the fused search kernel, which is the whole
    step, and the per-stage network kernels in the shapes the lock-step path launches them
    (`stored`: an engine that keeps every node's latent, which those kernels read)."""
    nets, G = mcts.nets, mcts.G
    out = {}
    out["k_search[fused MCTS.run]"] = dict(s=time_kernel(step, iters=20), flop=G * search_flop_per_tree(plan, mcts.S, mcts.depth),
                                           calls=1.0 / mcts.S)
    out["k_search[fused MCTS.run, stored latents]"] = dict(s=time_kernel(stored_step, iters=10),
                                                           flop=G * search_flop_per_tree(plan, mcts.S, mcts.depth), calls=0)
    mcts = stored
    pool = mcts.pool
    rows = (mcts.tree_base + 1).contiguous()
    rows_rep = rows.repeat_interleave(A)
    base = 1 + A
    next_rows = ((mcts.tree_base + base)[:, None] + mcts.arange_a[None, :]).reshape(-1).contiguous()
    out["k_dynamics[expand B=G*A]"] = dict(
        s=time_kernel(lambda: nets.dynamics(pool.latent, G * A, rows=rows_rep, next_latent=pool.latent,
                                            next_rows=next_rows, reward=mcts.reward)),
        flop=G * A * (2 * D * D + mlp_flop(plan.reward)), calls=0)
    out["k_prediction[policy B=G]"] = dict(
        s=time_kernel(lambda: nets.prediction(pool.latent, G, rows=rows, softmax_times=1, policy=mcts.prior)),
        flop=G * mlp_flop(plan.policy), calls=0)
    out["k_dynamics[step B=G]"] = dict(
        s=time_kernel(lambda: nets.dynamics(pool.latent, G, rows=rows, action=mcts.act, next_latent=mcts.tmp[0],
                                            reward=mcts.rollout_t[0])),
        flop=G * (2 * D * D + mlp_flop(plan.reward)), calls=0)
    out["k_prediction[value B=G]"] = dict(
        s=time_kernel(lambda: nets.prediction(mcts.tmp[0], G, value=mcts.rollout_t[1])),
        flop=G * mlp_flop(plan.value), calls=0)
    return out


def cpu_baseline(g, f, budget_s=12.0, seed=1):
    """The per-node CPU port (same Python/torch call structure as the reference) on one core."""
    from oracle import pyref
    torch.set_num_threads(1)
    rng = np.random.default_rng(seed)
    ev = pyref.TorchEvaluator(g, f, torch.arange(A), 1)
    sims = moves = 0
    with torch.no_grad():
        for warm in (True, False):
            t0 = time.perf_counter()
            while True:
                lat = torch.from_numpy(rng.standard_normal((1, 4, 6, 7)).astype(np.float32))
                pyref.run_mcts(lat, 1, ev, A, 50, discount=0.998, noise_frac=0.25)
                if warm:
                    break
                sims += 50
                moves += 1
                if time.perf_counter() - t0 > budget_s:
                    break
        dt = time.perf_counter() - t0
    return {"value": sims / dt, "unit": "sims/s", "cores": 1, "kind": "port",
            "sample": f"{moves} single-game MCTS.run calls (50 sims, c4_mlp nets, torch CPU 1 thread) in {dt:.1f}s",
            "env_steps_per_s": moves / dt}


def cpu_baseline_all_cores(budget_s=10.0):
    """The same port on P independent processes (P = min(16, cores): the GPU box gives one GPU's
    job 16 cores), each playing its own games -- how the reference would be scaled on a host.
    Must run BEFORE this process touches the GPU: the workers are started as child programs."""
    import subprocess
    procs = max(1, min(16, os.cpu_count() or 1))
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(100 + i), str(budget_s)],
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True) for i in range(procs)]
    sims = 0.0
    ok = 0
    for pr in ps:
        try:
            out, _ = pr.communicate(timeout=budget_s * 4 + 180)
            r = json.loads(out.strip().splitlines()[-1])
            sims += r["value"]
            ok += 1
        except Exception:
            pr.kill()
    if ok == 0:
        return None
    return {"value": sims, "unit": "sims/s", "cores": ok, "kind": "port",
            "sample": f"{ok} processes x {budget_s:.0f}s of single-game MCTS.run calls, 1 torch thread each"}


PMC_FILES = {"search": "r3_search_pmc.json", "search_stored": "r3_search_stored_pmc.json",
             "towers_256": "r3_towers_256_pmc.json", "towers_4096": "r3_towers_4096_pmc.json"}


def pmc_traffic(which, kernel):
    """(HBM bytes per launch of `kernel`, file) from the committed counter passes (tools/r3_profiles.sh: FETCH_SIZE
    and WRITE_SIZE need separate profiler runs, so they cannot be collected inside this process); (None, None) when
    there is no such record."""
    path = os.path.join(ROOT, "profiles", PMC_FILES.get(which, "-"))
    if not os.path.exists(path):
        return None, None
    with open(path) as fh:
        doc = json.load(fh)
    for name, e in doc["kernels"].items():
        if name.startswith(kernel) and "hbm_bytes_per_launch" in e:
            return e["hbm_bytes_per_launch"], "profiles/" + PMC_FILES[which]
    return None, None


def tower_search(G, S, dev, steps, with_roofline):
    """The lock-step search with the residual-tower g / f of the shipped YAMLs (latent
    [256, 6, 6], 5 actions; the networks of the CarRacing configuration): whole searches per
    second in the form self-play uses (bottleneck values per node, recorded launch graph), and
    the tower kernel's rollout-step call against the HBM roofline."""
    from muzero_amd.engine import BatchedMCTS
    from muzero_amd.towers import TowerNetworks
    from tools.bench_towers import build_nets
    shape, A = (256, 6, 6), 5
    D = shape[0] * shape[1] * shape[2]
    g, f = build_nets(shape, A)
    # roots in the form the shipped representation network produces them (src/config/config_car_racing.yaml:117-131,
    # neural_network.py:51,60-61): the final 1x1 convolution 3 -> 256 of 3 x 6 x 6 pooled values; the search starts
    # from those values (`root_t`), as self-play does
    torch.manual_seed(1)
    fin = torch.nn.Conv2d(3, shape[0], 1).to(dev)
    nets = TowerNetworks(g, f, device=dev, root_final=fin)
    mcts = BatchedMCTS(nets, G, S, 1, discount=0.998, noise_frac=0.25, device=dev, keep_latents=False)
    gen = torch.Generator(device=dev).manual_seed(0)
    root_t = torch.randn(G, 3, shape[1], shape[2], device=dev, generator=gen)
    with torch.no_grad():
        lat = fin(root_t).reshape(G, D).contiguous()
    root_kw = {"root_t": root_t} if mcts.root_form_ok(root_t) else {}
    eta = torch.distributions.Dirichlet(torch.full((A,), 0.3)).sample((G,)).float().to(dev)
    rand = torch.randint(0, A, (G, S + 1), device=dev, generator=gen, dtype=torch.int32)
    to_play = torch.zeros(G, dtype=torch.int8, device=dev)
    for _ in range(3):  # launch by launch, then the recording, then a replay
        res = mcts.run(lat, to_play, eta, rand, **root_kw)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        res = mcts.run(lat, to_play, eta, rand, **root_kw)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    assert int(res.visits.sum().item()) == G * S and mcts.replays >= steps
    out = {"workload": f"{G} trees x {S} sims, residual-tower g/f, latent {list(shape)}, {A} actions, lock-step search, roots as "
                       + ("the bottleneck values of h's final convolution" if root_kw else "full latents"),
           "sims_per_sec": G * S / dt, "ms_per_search": dt * 1e3, "steps": steps}
    if with_roofline and getattr(nets, "fused_g", False):
        # the kernel that takes most of a step: fc_merge on the bottleneck values + the first product of g's tower
        # for an expansion (every tree's leaf with all A actions); fp32 matrix instructions
        t = torch.randn(G, nets.bottleneck_g, device=dev)
        s = time_kernel(lambda: nets.merge_conv(t, None, None, G), iters=20)
        hw = shape[1] * shape[2]
        flop = 2.0 * G * nets.merge_t_perm.Kp * D + 2.0 * G * A * hw * shape[0] * 32
        from tools.bench_towers import merge_conv_roofline
        out["roofline_tower"] = merge_conv_roofline(nets, G, A, D, hw, shape[0], s, flop)
        # the expansion call's launches (A actions per tree) dominate that kernel's record
        out["roofline_tower"]["traffic"], out["roofline_tower"]["traffic_source"] = pmc_traffic(f"towers_{G}", "k_merge_conv_bf")
        if out["roofline_tower"]["traffic_source"]:
            out["roofline_tower"]["traffic_source"] += (" (mean over the kernel's launches in a search: expansion calls, "
                                                        "as timed here, alternate with rollout-step calls of 1/A the rows)")
    del mcts, nets
    torch.cuda.empty_cache()
    return out


def representation_time(G, dev, iters=3):
    """The shipped convolutional h (32-frame history of 3x96x96 -> latent [256, 6, 6]) for G games
    in one call on the HIP convolution path: milliseconds per self-play move."""
    from muzero_amd.conv_repr import ConvRepresentation
    from tools.bench_conv_repr import flops_per_game, shipped_h
    run = ConvRepresentation(shipped_h(), device=dev)
    x = torch.rand(G, 128, 96, 96, device=dev) * 255.0
    run(x)
    s = time_kernel(lambda: run(x), iters=iters)
    del run, x
    torch.cuda.empty_cache()
    tf = flops_per_game() * G / s / 1e12
    return {"games": G, "ms_per_move": s * 1e3, "tflops": tf, "fp32_mfma_peak_tflops": FP32_MFMA_PEAK_TFLOPS,
            # every float32 product is three fp16 products on the matrix cores (whole network, small layers included)
            "f16_executed_tflops": 3 * tf, "f16_mfma_peak_tflops": F16_MFMA_PEAK_TFLOPS,
            "f16_executed_frac": 3 * tf / F16_MFMA_PEAK_TFLOPS,
            "kernel": "convolutions as three fp16 matrix products on two-plane operands, float32 accuracy "
                      "(k_conv3_same / k_conv3), per-game BatchNorm statistics from the convolution epilogue"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--games", type=int, default=4096, help="concurrent games per GPU")
    ap.add_argument("--sims", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--store-latents", action="store_true",
                    help="keep every node's latent in the pool (the drop-in MCTS.run's form) instead of one merge row per "
                         "expansion (self-play's form; bit-identical trees)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--selfplay-moves", type=int, default=64, help="timed self-play moves per game (0 = skip)")
    ap.add_argument("--min-seconds", type=float, default=6.0,
                    help="after the K timed steps, keep stepping for this long (a sustained-rate figure; longer than a "
                         "5-second utilisation sampler's period)")
    ap.add_argument("--more-games", type=int, nargs="*", default=[8192, 16384],
                    help="also time the search at these game counts per GPU (secondary figures; empty = skip)")
    ap.add_argument("--tower-games", type=int, nargs="*", default=[256, 4096],
                    help="also time the residual-tower search at these tree counts (empty = skip)")
    ap.add_argument("--cpu-worker", nargs=2, metavar=("SEED", "SECONDS"), help="internal: one CPU-baseline process")
    args = ap.parse_args()
    if args.cpu_worker:  # a child of cpu_baseline_all_cores: CPU only, prints one JSON line
        _, _, g, f = build_networks()
        print(json.dumps(cpu_baseline(g, f, float(args.cpu_worker[1]), int(args.cpu_worker[0]))))
        return
    multi = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu_baseline:
        multi = cpu_baseline_all_cores()  # child programs: started before the GPU is initialised

    from muzero_amd import sharding
    shard = sharding.shard_from_env(args.games)
    rank, world, local = shard.rank, shard.world, shard.local_rank
    # one rank per GPU; MZ_DIST_BACKEND=gloo lets several ranks rehearse the N > 1 path on one card
    backend = os.environ.get("MZ_DIST_BACKEND", "nccl")
    local_dev = local if backend == "nccl" else local % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run: one rank per GPU over RCCL
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from muzero_amd.engine import BatchedMCTS, PackedNetworks
    cfg, h, g, f = build_networks()
    G, S = args.games, args.sims
    nets = PackedNetworks(g, f, device=dev)
    mcts = BatchedMCTS(nets, G, S, 1, discount=0.998, noise_frac=0.25, device=dev, keep_latents=args.store_latents)
    latent, to_play = synthetic_roots(h, G, seed=100 + rank, device=dev)
    eta = torch.from_numpy(np.random.default_rng(7 + rank).dirichlet(np.full(A, 0.3), size=G).astype(np.float32)).to(dev)
    rand = torch.from_numpy(child_choice_streams(G, S, base_seed=shard.first_game)).to(dev)

    def step():
        return mcts.run(latent, to_play, eta, rand)

    def fence():
        sharding.fence(dist)

    # set-up, not part of the measurement: the first call records the launch graph, and a few hundred
    # milliseconds of the same step bring the chip from its idle clocks to the state a running job is in
    step()
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.5:
        step()
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        step()
    dt = sharding.timed_steps(step, args.steps, dist, device=dev)  # barrier + sync both sides, max over ranks
    res = mcts.finish()
    assert int(res.visits.sum().item()) == G * S, "search did not complete every simulation"

    line = {
        "metric": "mcts_simulations_per_sec", "value": sharding.whole_job_rate(G * S, args.steps, world, dt),
        "unit": "sims/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (products as 3 fp16 matrix products on two-plane operands, f32 accumulate) + f64 tree", "data": "synthetic",
        "config": {"workload": f"c4_mlp: {G} Connect-Four games per GPU x {S} sims/move, A=7, D=168, "
                               "model_look_ahead=1, noise_frac=0.25 (fixed eta), MT19937 child-choice streams",
                   "games_per_gpu": G, "sims_per_move": S, "parallelism": f"games sharded x{world}, no collective",
                   "latents": "every node's latent stored" if args.store_latents else
                              "one merge row per expansion; a child's latent is relu(row + table[action]) when read (self-play's "
                              "form; trees, policies and values bit-identical to the stored form)"},
        "searches_per_sec": world * G * args.steps / dt,
    }
    if dist is not None:
        # every rank reports in: an N > 1 record shows that RCCL really connected N ranks
        seen = torch.ones(1, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(seen)
        line["rccl_ranks"] = int(seen.item())
        line["dist_backend"] = backend
    if args.min_seconds > 0:
        # the K timed steps above are tens of milliseconds; a sustained region of the same step lets a
        # sampling monitor see the GPU busy and shows the rate holds under a warm chip
        n_sus = max(args.steps, int(args.min_seconds / max(dt / args.steps, 1e-6)) + 1)
        dts = sharding.timed_steps(step, n_sus, dist, device=dev)
        line["sustained"] = {"steps": n_sus, "seconds": dts, "value": sharding.whole_job_rate(G * S, n_sus, world, dts),
                             "ms_per_step": dts / n_sus * 1e3}
    if args.selfplay_moves > 0:
        # whole self-play moves: observe -> history -> h -> MCTS.run -> action pick -> env step ->
        # restart of finished games, for G games per GPU; the per-move Dirichlet, child-choice and
        # exploration streams are drawn on the device
        from muzero_amd.selfplay import SelfPlayEngine
        for graphed in (False, True):
            # launch by launch (some twenty-five launches per move), then the whole move replayed from one HIP graph
            sp = SelfPlayEngine(h, g, f, G, cfg.training_data_generator, device=dev, seed=rank, device_rng=True, graph_moves=graphed)
            for _ in range(4):
                sp.move()
            fence()
            t0 = time.perf_counter()
            for _ in range(args.selfplay_moves):
                sp.move()
            fence()
            dts = sharding.max_over_ranks(time.perf_counter() - t0, dist, dev)
            key = "" if graphed else "_launch_by_launch"
            line["selfplay_env_steps_per_sec" + key] = world * G * args.selfplay_moves / dts
            line["selfplay_ms_per_move" + key] = dts / args.selfplay_moves * 1e3
            del sp
    if rank == 0 and not args.no_roofline:
        stored = mcts if args.store_latents else BatchedMCTS(nets, G, S, 1, discount=0.998, noise_frac=0.25, device=dev,
                                                             keep_latents=True)
        ks = kernel_rooflines(mcts, nets.plan, step, stored, lambda: stored.run(latent, to_play, eta, rand))
        del stored
        dom = "k_search[fused MCTS.run]"
        # `achieved`: ALGORITHMIC flops of one launch (the reference's formulation x G*S simulations) over the
        # kernel's measured duration.  The kernel computes every float32 product as three fp16 matrix products
        # (operands as two fp16 planes, float32 accumulation: float32-accurate results), so the rate such
        # arithmetic can reach is a third of the fp16 pipe's peak: that is `peak`.  What the pipe really runs
        # (fewer products than the reference formulation: discarded heads skipped, the merge layer split into
        # W[:, :D] x + a per-action table; more flops per product: three, on padded tiles) is `executed_*`.
        alg = G * S * reference_flop_per_sim(nets.plan, mcts.depth)
        exe = G * search_f16_flop_per_tree(nets.plan, S, mcts.depth)
        tf_alg = alg / ks[dom]["s"] / 1e12
        tf_exe = exe / ks[dom]["s"] / 1e12
        peak = F16_MFMA_PEAK_TFLOPS / F16_PRODUCTS_PER_FP32_PRODUCT
        traffic, traffic_file = pmc_traffic("search_stored" if args.store_latents else "search", "k_search") \
            if (G, S) == (4096, 50) else (None, None)
        lat_bytes = LATENT_BYTES_PER_SIM if args.store_latents else IMPLICIT_LATENT_BYTES_PER_SIM
        line["roofline"] = {"schema": ROOFLINE_SCHEMA, "bound": "mfma", "kernel": dom, "achieved": tf_alg, "peak": peak,
                            "unit": "TFLOP/s", "frac": tf_alg / peak,
                            "peak_note": "dense fp16 matrix peak 2500 TFLOP/s / 3 fp16 products per float32-accurate product",
                            "traffic": traffic,
                            "traffic_source": f"committed profile ({traffic_file}), not measured in this run",
                            "avg_launch_us": ks[dom]["s"] * 1e6, "algorithmic_flop_per_launch": alg,
                            "algorithmic_tflops": tf_alg, "fp32_mfma_peak_tflops": FP32_MFMA_PEAK_TFLOPS,
                            "fp32_equivalent_frac": tf_alg / FP32_MFMA_PEAK_TFLOPS,
                            "executed_f16_flop_per_launch": exe, "executed_f16_tflops": tf_exe,
                            "f16_mfma_peak_tflops": F16_MFMA_PEAK_TFLOPS, "executed_frac": tf_exe / F16_MFMA_PEAK_TFLOPS,
                            "algorithmic_hbm_GBps": G * S * (TREE_BYTES_PER_SIM + lat_bytes) / ks[dom]["s"] / 1e9,
                            "hbm_peak_GBps": 8000.0}
        line["kernels"] = {k: {"avg_us": v["s"] * 1e6, "tflops": v["flop"] / v["s"] / 1e12} for k, v in ks.items()}
        if not args.store_latents:  # the same search with every node's latent written to the pool: the drop-in MCTS.run's form
            other = ks["k_search[fused MCTS.run, stored latents]"]["s"]
            line["stored_latent_form"] = {"sims_per_sec": G * S / other, "ms_per_step": other * 1e3}
    if rank == 0 and args.more_games:
        # secondary figures: more games per GPU (two or four 16-tree workgroups per compute unit)
        line["more_games"] = {}
        for mg in args.more_games:
            m2 = BatchedMCTS(nets, mg, S, 1, discount=0.998, noise_frac=0.25, device=dev, keep_latents=args.store_latents)
            reps = (mg + G - 1) // G
            lat2, tp2 = latent.repeat(reps, 1)[:mg].contiguous(), to_play.repeat(reps)[:mg].contiguous()
            eta2, rand2 = eta.repeat(reps, 1)[:mg].contiguous(), rand.repeat(reps, 1)[:mg].contiguous()
            s2 = time_kernel(lambda: m2.run(lat2, tp2, eta2, rand2), iters=10)
            assert int(m2.finish().visits.sum().item()) == mg * S
            line["more_games"][str(mg)] = {"sims_per_sec": mg * S / s2, "ms_per_step": s2 * 1e3}
            del m2, lat2, tp2, eta2, rand2
            torch.cuda.empty_cache()
    if rank == 0 and args.tower_games:
        del mcts
        torch.cuda.empty_cache()
        line["towers"] = {}
        for tg in args.tower_games:
            r = tower_search(tg, S, dev, steps=5 if tg <= 1024 else 3, with_roofline=not args.no_roofline)
            line["towers"][str(tg)] = r
        line["towers"]["representation"] = representation_time(256, dev)
        from tools.bench_conv_repr import flappy_time
        line["towers"]["representation_flappy"] = flappy_time(16)  # FlappyBird's h: 3x512x288 frames, 16 games per call
        # whole self-play moves with host-side simulators (stand-ins returning random frames) and the shipped network
        # shapes: the serial engine, and two halves alternating between the host thread pool and the GPU
        from tools.bench_selfplay_towers import run as selfplay_host_envs
        line["towers"]["selfplay_host_envs"] = selfplay_host_envs(256, moves=8, parts=False)
        first = line["towers"][str(args.tower_games[0])]
        line["towers"].update(sims_per_sec=first["sims_per_sec"], ms_per_search=first["ms_per_search"],
                              roofline_tower=first.get("roofline_tower"))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(g, f)
        line["cpu_baseline"]["all_cores"] = multi
        line["cpu_baseline"]["note"] = ("the per-node CPU port (oracle/pyref.py: the reference's Python/torch call structure, "
                                        "pinned bit for bit by the reference's trees), not the reference itself, which "
                                        "cannot run on the GPU box; in the build container the imported reference ran "
                                        "866 sims/s on one core on this configuration, the port about 4x that")
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line))


if __name__ == "__main__":
    main()
