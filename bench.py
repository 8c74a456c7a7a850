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
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X dense fp32 matrix peak
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
LATENT_BYTES_PER_SIM = 4 * D * (A + 2)      # read the leaf, write A children, read the drawn child


def kernel_rooflines(mcts, plan, step):
    """Times (HIP events on the launch stream) the fused search kernel, which is the whole
    step, and the per-stage network kernels in the shapes the lock-step path launches them."""
    nets, pool, G = mcts.nets, mcts.pool, mcts.G
    out = {}
    out["k_search[fused MCTS.run]"] = dict(s=time_kernel(step, iters=5), flop=G * search_flop_per_tree(plan, mcts.S, mcts.depth),
                                           calls=1.0 / mcts.S)
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


def pmc_traffic(G, S):
    """HBM bytes per launch of the search kernel from the committed counter passes (FETCH_SIZE and
    WRITE_SIZE need separate profiler runs, so they cannot be collected inside this process);
    None when the workload differs from the profiled one."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1_f_pmc_traffic.json")
    if not os.path.exists(path) or (G, S) != (4096, 50):
        return None
    with open(path) as fh:
        return json.load(fh)["hbm_bytes_per_launch"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--games", type=int, default=4096, help="concurrent games per GPU")
    ap.add_argument("--sims", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--selfplay-moves", type=int, default=8, help="timed self-play moves per game (0 = skip)")
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
    mcts = BatchedMCTS(nets, G, S, 1, discount=0.998, noise_frac=0.25, device=dev)
    latent, to_play = synthetic_roots(h, G, seed=100 + rank, device=dev)
    eta = torch.from_numpy(np.random.default_rng(7 + rank).dirichlet(np.full(A, 0.3), size=G).astype(np.float32)).to(dev)
    rand = torch.from_numpy(child_choice_streams(G, S, base_seed=shard.first_game)).to(dev)

    def step():
        return mcts.run(latent, to_play, eta, rand)

    def fence():
        sharding.fence(dist)

    for _ in range(args.warmup):
        step()
    dt = sharding.timed_steps(step, args.steps, dist, device=dev)  # barrier + sync both sides, max over ranks
    res = mcts.finish()
    assert int(res.visits.sum().item()) == G * S, "search did not complete every simulation"

    line = {
        "metric": "mcts_simulations_per_sec", "value": sharding.whole_job_rate(G * S, args.steps, world, dt),
        "unit": "sims/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32+f64", "data": "synthetic",
        "config": {"workload": f"c4_mlp: {G} Connect-Four games per GPU x {S} sims/move, A=7, D=168, "
                               "model_look_ahead=1, noise_frac=0.25 (fixed eta), MT19937 child-choice streams",
                   "games_per_gpu": G, "sims_per_move": S, "parallelism": f"games sharded x{world}, no collective"},
        "searches_per_sec": world * G * args.steps / dt,
    }
    if args.selfplay_moves > 0:
        # whole self-play moves: observe -> history -> h -> MCTS.run -> action pick -> env step ->
        # restart of finished games, for G games per GPU; the per-move Dirichlet, child-choice and
        # exploration streams are drawn on the device
        from muzero_amd.selfplay import SelfPlayEngine
        sp = SelfPlayEngine(h, g, f, G, cfg.training_data_generator, device=dev, seed=rank, device_rng=True)
        for _ in range(2):
            sp.move()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.selfplay_moves):
            sp.move()
        fence()
        dts = sharding.max_over_ranks(time.perf_counter() - t0, dist, dev)
        line["selfplay_env_steps_per_sec"] = world * G * args.selfplay_moves / dts
        line["selfplay_ms_per_move"] = dts / args.selfplay_moves * 1e3
    if rank == 0 and not args.no_roofline:
        ks = kernel_rooflines(mcts, nets.plan, step)
        dom = "k_search[fused MCTS.run]"
        # `achieved`: algorithmic flops of one launch (reference formulation x G*S simulations)
        # over the kernel's measured duration.  The kernel EXECUTES fewer flops for the same
        # results (heads whose output the search discards are skipped, the merge layer is split
        # into W[:, :D] x + a per-action table): `executed_*` is what the MFMA pipe really does.
        alg = G * S * reference_flop_per_sim(nets.plan, mcts.depth)
        tf_alg = alg / ks[dom]["s"] / 1e12
        tf_exe = ks[dom]["flop"] / ks[dom]["s"] / 1e12
        line["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": tf_alg, "peak": FP32_MFMA_PEAK_TFLOPS,
                            "unit": "TFLOP/s", "frac": tf_alg / FP32_MFMA_PEAK_TFLOPS, "traffic": pmc_traffic(G, S),
                            "avg_launch_us": ks[dom]["s"] * 1e6, "algorithmic_flop_per_launch": alg,
                            "executed_flop_per_launch": ks[dom]["flop"], "executed_tflops": tf_exe,
                            "executed_frac": tf_exe / FP32_MFMA_PEAK_TFLOPS,
                            "algorithmic_hbm_GBps": G * S * (TREE_BYTES_PER_SIM + LATENT_BYTES_PER_SIM) / ks[dom]["s"] / 1e9,
                            "hbm_peak_GBps": 8000.0}
        line["kernels"] = {k: {"avg_us": v["s"] * 1e6, "tflops": v["flop"] / v["s"] / 1e12} for k, v in ks.items()}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(g, f)
        line["cpu_baseline"]["all_cores"] = multi
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line))


if __name__ == "__main__":
    main()
