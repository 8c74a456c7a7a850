"""Two (or more) ranks of DataParallelLearner.step_local_recorded on ONE card over gloo: a rehearsal of the
multi-GPU learner step (three graph segments, two gradient all-reduces issued between them) where only one GPU
is at hand.  Each rank takes its share of every batch; rank r writes losses and final weights to <out>/rec<r>.pt.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \\
        tools/rehearse_recorded_dp.py <out_dir>
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def batches(n, B, seed):
    from muzero_amd.learner import WindowBatch
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        hist = torch.from_numpy(rng.integers(0, 2, size=(B, 3, 6, 7)).astype(np.float32))
        acts = torch.from_numpy(rng.integers(0, 7, size=(B, 6)))
        pols = torch.from_numpy(rng.dirichlet(np.ones(7), size=(B, 6)).astype(np.float32))
        z = torch.from_numpy(rng.normal(size=(B, 6)).astype(np.float32))
        r = torch.from_numpy(rng.normal(size=(B, 6)).astype(np.float32))
        w = torch.from_numpy(rng.uniform(0.2, 1.0, size=B).astype(np.float32))
        w[rng.integers(0, B)] = 0.0  # a zero-weight window, as the prioritised sampler pads with
        out.append(WindowBatch(hist, acts, pols, z, r, w).to("cuda"))
    return out


def networks():
    from muzero_amd.config import load_config
    from muzero_amd.neural_networks import DynamicsNetwork, PredictionNetwork, RepresentationNetwork
    cfg = load_config("config_c4_mlp.yaml")
    torch.manual_seed(0)
    h = RepresentationNetwork((2, 6, 7), cfg.networks.latent_shape, cfg.networks.representation)
    g = DynamicsNetwork(cfg.networks.latent_shape, 7, cfg.networks.dynamics)
    f = PredictionNetwork(cfg.networks.latent_shape, 7, cfg.networks.prediction)
    return [n.cuda() for n in (h, g, f)]


def main() -> None:
    import torch.distributed as dist
    from muzero_amd.learner import DataParallelLearner
    out = sys.argv[1]
    if os.environ.get("MZ_REHEARSE_WATCHDOG"):  # a hung collective: print where every thread stands, then leave
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["MZ_REHEARSE_WATCHDOG"]), exit=True)
    os.makedirs(out, exist_ok=True)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    try:
        torch.cuda.set_device(0)
        nets = networks()
        params = [p for n in nets for p in n.parameters()]
        dp = DataParallelLearner(*nets, torch.optim.Adam(params, lr=1e-3, capturable=True), 5, (1.0, 0.5, 2.0), dist=dist)
        losses = []
        for b in batches(3, 24, seed=21):
            mine = torch.arange(rank, len(b), world, device="cuda")
            loss, per_window = dp.step_local_recorded(b.rows(mine))
            torch.cuda.synchronize()
            assert per_window.shape == (len(mine),)
            losses.append(float(loss))
        torch.save({"losses": losses, "params": [p.detach().cpu().clone() for p in params]}, os.path.join(out, f"rec{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()
    print(f"rank {rank}: losses {losses}", flush=True)


if __name__ == "__main__":
    main()
