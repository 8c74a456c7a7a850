"""Actor-learner loop on one node, one process per GPU (launch with torch.distributed.run for
more than one): every rank plays its own block of Connect-Four games with the fused search,
records the moves into its device episode store, cuts training windows out of it and takes part
in data-parallel learner steps (one flat gradient all-reduce per step over RCCL); after every
learner phase the rank re-packs the new weights for its searches.

    python tools/actor_learner_loop.py --games 4096 --iters 3
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/actor_learner_loop.py

Prints one JSON line (rank 0): self-play env steps/s, learner steps/s, loss per iteration.
Reference flow: main.py:97-154 (generate_train_model_loop)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=4096, help="concurrent games per GPU")
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--moves-per-iter", type=int, default=42)
    ap.add_argument("--steps-per-iter", type=int, default=16)
    ap.add_argument("--batch", type=int, default=256, help="windows per rank and learner step")
    ap.add_argument("--recorded", action="store_true",
                    help="learner steps replayed from HIP graphs (DataParallelLearner.step_local_recorded)")
    ap.add_argument("--tune-gemms", action="store_true",
                    help="torch times the BLAS solutions for the learner's matrix products once (untimed warm-up) "
                         "and keeps the fastest (muzero_amd.learner.tune_small_gemms)")
    ap.add_argument("--no-move-graph", action="store_true", help="issue every self-play move launch by launch")
    ap.add_argument("--prioritized", action="store_true",
                    help="episodes drawn by priority (alpha 0.6, beta 1.0, src/replay_buffer.py:11-39 defaults), "
                         "priorities updated from the per-window loss; default: uniform")
    args = ap.parse_args()

    import bench
    from muzero_amd import sharding
    from muzero_amd.episode_store import DeviceEpisodeStore
    from muzero_amd.learner import DataParallelLearner, tune_small_gemms
    from muzero_amd.selfplay import SelfPlayEngine
    from muzero_amd.training import z_targets_device

    shard = sharding.shard_from_env(args.games)
    torch.cuda.set_device(shard.local_rank)
    dev = torch.device("cuda", shard.local_rank)
    dist = None
    if shard.world > 1 or "RANK" in os.environ:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    if args.tune_gemms:
        tune_small_gemms()
    cfg, h, g, f = bench.build_networks()  # identical seed-0 weights on every rank
    h, g, f = h.to(dev), g.to(dev), f.to(dev)
    tcfg, gcfg = cfg.training, cfg.training_data_generator
    eng = SelfPlayEngine(h, g, f, args.games, gcfg, device=dev, seed=shard.rank, device_rng=True, graph_moves=not args.no_move_graph)
    store = DeviceEpisodeStore(args.games, gcfg.max_steps_per_episode, (2, 6, 7), 7, capacity=4 * args.games, device=dev)
    params = [p for n in (h, g, f) for p in n.parameters()]
    opt = torch.optim.Adam(params, lr=tcfg.learning_rate, betas=tuple(tcfg.betas), weight_decay=tcfg.weight_decay,
                           capturable=args.recorded)
    learner = DataParallelLearner(h, g, f, opt, tcfg.roll_ahead,
                                  (tcfg.policy_coefficient, tcfg.value_coefficient, tcfg.reward_coefficient), dist=dist)
    gen = torch.Generator(device=dev).manual_seed(1000 + shard.rank)
    learner_step = learner.step_local_recorded if args.recorded else learner.step_local

    # untimed warm-up: first uses of the torch operators load their code objects
    for _ in range(3):
        store.record(*eng.move())
    for _ in range(gcfg.max_steps_per_episode):
        store.record(*eng.move())

    def one_step():
        if args.prioritized:
            batch, slots, _ = store.sample_prioritized(args.batch, tcfg.look_back, tcfg.roll_ahead, h.history_length,
                                                       tcfg.discount_factor, z_targets_device, alpha=0.6, beta=1.0,
                                                       generator=gen)
            loss, errors = learner_step(batch)
            store.update_priorities(slots, errors)
        else:
            batch, _, _ = store.sample_windows(args.batch, tcfg.look_back, tcfg.roll_ahead, h.history_length,
                                               tcfg.discount_factor, z_targets_device, generator=gen)
            loss, _ = learner_step(batch)
        return loss

    for _ in range(3):  # and of the sampler / autograd / the optimizer
        one_step()
    eng.refresh()

    t_play = t_learn = 0.0
    moves = steps = 0
    losses = []
    for it in range(args.iters):
        sharding.fence(dist)
        t0 = time.perf_counter()
        for _ in range(args.moves_per_iter):
            store.record(*eng.move(training_steps=steps))
        sharding.fence(dist)
        t_play += sharding.max_over_ranks(time.perf_counter() - t0, dist, dev)
        moves += args.moves_per_iter
        t0 = time.perf_counter()
        last = float("nan")
        for _ in range(args.steps_per_iter):
            last = one_step()
            steps += 1
        eng.refresh()  # the actors search with the new weights from the next move on
        sharding.fence(dist)
        t_learn += sharding.max_over_ranks(time.perf_counter() - t0, dist, dev)
        losses.append(float(last))
    if shard.rank == 0:
        print(json.dumps({
            "n_gpus": shard.world, "games_per_gpu": args.games, "iters": args.iters,
            "selfplay_env_steps_per_sec": shard.world * args.games * moves / t_play,
            "learner_steps_per_sec": steps / t_learn, "windows_per_step": shard.world * args.batch, "prioritized": args.prioritized, "recorded": args.recorded, "tuned_gemms": args.tune_gemms,
            "episodes_stored_rank0": store.n_finished, "loss_per_iter": losses,
            "seconds": {"selfplay": t_play, "learner": t_learn}}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
