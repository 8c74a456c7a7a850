"""The learner step replayed from HIP graphs (DataParallelLearner.step_local_recorded) takes the
steps DataParallelLearner.step_local takes: same losses, same weights after several steps with
different batches, with Adam and with plain SGD; recording leaves weights and optimizer state
untouched."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _batches(n, B, seed):
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


@pytest.mark.parametrize("opt_name", ["adam", "sgd"])
def test_recorded_step_equals_eager_step(opt_name):
    from muzero_amd.learner import DataParallelLearner
    from tests.test_gpu_nets import _c4_nets
    nets_a = [n.cuda() for n in _c4_nets(0)]
    nets_b = copy.deepcopy(nets_a)

    def make(nets):
        params = [p for n in nets for p in n.parameters()]
        if opt_name == "adam":
            opt = torch.optim.Adam(params, lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-4, capturable=True)
        else:
            opt = torch.optim.SGD(params, lr=1e-2)
        return DataParallelLearner(*nets, opt, 5, (1.0, 0.5, 2.0)), params

    eager, pa = make(nets_a)
    rec, pb = make(nets_b)
    batches = _batches(4, 32, seed=7)
    before = [p.detach().clone() for p in pb]
    rec._record(batches[0], None)  # recording alone changes nothing
    for p, q in zip(pb, before):
        assert torch.equal(p, q)
    if opt_name == "adam":
        assert all(float(st["step"]) == 0 and float(st["exp_avg"].abs().max()) == 0 for st in rec.optimizer.state.values())
    rec._rec_shapes = None
    for b in batches:
        loss_e, per_e = eager.step_local(b)
        loss_r, per_r = rec.step_local_recorded(b)
        torch.cuda.synchronize()
        assert abs(float(loss_r) - loss_e) <= 1e-6 * abs(loss_e)
        torch.testing.assert_close(per_r.cpu(), per_e, rtol=1e-6, atol=1e-7)
    # Adam divides by sqrt(v): an element whose gradient is small beside its summands carries the
    # BLAS summation-order difference between capture and eager execution at the scale of a step
    # (lr = 1e-3), so its bound is 1e-3 of one step; SGD keeps the tight bound
    atol = 1e-6 if opt_name == "adam" else 1e-8
    for p, q in zip(pa, pb):
        torch.testing.assert_close(p, q, rtol=1e-6, atol=atol)
    assert not torch.equal(pb[0], before[0])
    # a batch of zero-weight windows only (every draw was an episode too short for a window): the
    # recorded step leaves finite, unchanged weights, as the eager step (which skips the update) does
    from muzero_amd.learner import WindowBatch
    dead = batches[0]
    dead = WindowBatch(dead.history, dead.actions, dead.policies, dead.z, dead.rewards, torch.zeros_like(dead.is_weights))
    kept = [p.detach().clone() for p in pb]
    loss_r, _ = rec.step_local_recorded(dead)
    loss_e, _ = eager.step_local(dead)
    torch.cuda.synchronize()
    assert float(loss_r) == 0.0 and loss_e == 0.0
    for p, q in zip(pb, kept):
        assert torch.isfinite(p).all() and torch.equal(p, q)


def test_batched_learner_step_reproduces_the_reference_learner():
    """The batched step (one unroll over all windows, fused loss kernel, SGD) against the run of the
    REFERENCE's NeuralNetworkManager.train recorded in tests/golden/selfplay_learner.npz
    (src/training.py:110-208): same sampled episodes and window starts (numpy / random seeded as the
    reference's run), so the same losses, priorities and trained weights -- eagerly and with the step
    replayed from HIP graphs."""
    import os
    import random
    from muzero_amd.config import TrainingConfig
    from muzero_amd.learner import DataParallelLearner, windows_to_batch
    from muzero_amd.replay_buffer import ReplayBuffer
    from muzero_amd.training import compute_z_targets_batch
    from muzero_amd.training_data_generator import Chunk, Episode
    from tests.conftest import GOLDEN
    from tests.test_gpu_dropin import _c4_nets_k2
    z = np.load(os.path.join(GOLDEN, "selfplay_learner.npz"))
    episodes = []
    for e in range(int(z["n_episodes"])):
        n = z[f"ep{e}_action"].shape[0]
        episodes.append(Episode(chunks=[Chunk(
            state=torch.from_numpy(z[f"ep{e}_state"][i:i + 1]), policy=torch.from_numpy(z[f"ep{e}_policy"][i]),
            reward=float(z[f"ep{e}_reward"][i]), value=float(z[f"ep{e}_value"][i]), best_action=int(z[f"ep{e}_action"][i]))
            for i in range(n)]))
    lookback, w, gamma = 2, 2, 0.9
    for recorded in (False, True):
        h, g, f = (n.cuda() for n in _c4_nets_k2())
        params = [p for n in (h, g, f) for p in n.parameters()]
        opt = torch.optim.SGD(params, lr=0.01, weight_decay=1e-4, momentum=0.9)
        dp = DataParallelLearner(h, g, f, opt, w, coeff=(0.25, 0.5, 1.0))   # policy, value, reward coefficients
        random.seed(9)
        np.random.seed(9)
        rb = ReplayBuffer(buffer_size=8, batch_size=3, alpha=0.6, beta=1.0)
        rb.add_episodes(episodes)
        losses = []
        for _ in range(3):   # mini_batch_size optimizer steps, as NeuralNetworkManager.train takes them
            batch, indices, is_w = rb.sample_batch()
            wins = [(ep, random.randrange(lookback, len(ep.chunks) - (w + 1) + 1)) for ep in batch]
            R = np.array([[float(ep.chunks[i].reward) for i in range(k, k + w + 1)] for ep, k in wins], np.float64)
            V = np.array([[float(ep.chunks[i].value) for i in range(k, k + w + 1)] for ep, k in wins], np.float64)
            wb = windows_to_batch(wins, compute_z_targets_batch(R, V, gamma), R, is_w, h.history_length, lookback, w).to("cuda")
            if recorded:
                loss, per_window = dp.step_local_recorded(wb)
                loss, per_window = float(loss), per_window.cpu()
            else:
                loss, per_window = dp.step_local(wb)
            losses.append(loss)
            rb.update_priorities(indices, [float(e) for e in per_window])
        np.testing.assert_allclose(losses, z["learner_losses"], rtol=1e-5)
        np.testing.assert_allclose([p for _, p in rb.episode_buffer], z["learner_priorities"], rtol=1e-5)
        np.testing.assert_allclose(g.fc_merge.bias.detach().cpu().numpy(), z["learner_fc_merge_bias"], rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(f.policy_mlp[-2].weight.detach().cpu().numpy(), z["learner_policy_last_weight"], rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(h.final_conv.weight.detach().cpu().numpy(), z["learner_repr_weight"], rtol=1e-4, atol=1e-7)


def test_fused_loss_kernel_matches_torch_losses_and_gradients():
    """mz_unroll_loss against F.cross_entropy / squared errors and autograd on the same predictions."""
    import torch.nn.functional as F
    from muzero_amd.learner import _FusedUnrollLoss
    torch.manual_seed(3)
    B, w, A = 300, 4, 7
    pp = torch.softmax(torch.randn(w + 1, B, A, device="cuda"), dim=-1).requires_grad_()
    pv = torch.randn(w + 1, B, device="cuda", requires_grad=True)
    pr = torch.randn(w, B, device="cuda", requires_grad=True)
    tp = torch.softmax(torch.randn(B, w + 1, A, device="cuda"), dim=-1)
    zt, rt = torch.randn(B, w + 1, device="cuda"), torch.randn(B, w + 1, device="cuda")
    isw = torch.rand(B, device="cuda")
    isw[::7] = 0.0
    cp, cv, cr = 0.25, 0.5, 2.0
    lp = sum(F.cross_entropy(pp[i], tp[:, i], reduction="none") for i in range(w + 1)) * cp
    lv = sum((pv[i] - zt[:, i]) ** 2 for i in range(w + 1)) * cv
    lr = sum((pr[i] - rt[:, i]) ** 2 for i in range(w)) * cr
    want_pw = lp + lv + lr
    want = (isw * want_pw).sum()
    want_grads = torch.autograd.grad(want * 3.0, (pp, pv, pr))
    got, pw, sums = _FusedUnrollLoss.apply(pp, pv, pr, tp, zt, rt, isw, cp, cv, cr)
    got_grads = torch.autograd.grad(got * 3.0, (pp, pv, pr))
    torch.testing.assert_close(pw, want_pw.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(got.detach(), want.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(sums[1:].detach(), torch.stack([lp.sum(), lv.sum(), lr.sum()]).detach(), rtol=1e-5, atol=1e-4)
    for a, b in zip(got_grads, want_grads):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)


def _recorded_dp_worker(rank, world, port, tmp):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)   # two ranks on the one card: gloo moves device tensors
    try:
        from muzero_amd.learner import DataParallelLearner
        from tests.test_gpu_nets import _c4_nets
        torch.cuda.set_device(0)
        nets = [n.cuda() for n in _c4_nets(0)]
        params = [p for n in nets for p in n.parameters()]
        opt = torch.optim.Adam(params, lr=1e-3, capturable=True)
        dp = DataParallelLearner(*nets, opt, 5, (1.0, 0.5, 2.0), dist=dist)
        losses = []
        for b in _batches(3, 24, seed=21):
            mine = torch.arange(rank, len(b), world, device="cuda")
            loss, per_window = dp.step_local_recorded(b.rows(mine))
            torch.cuda.synchronize()
            assert per_window.shape == (len(mine),)
            losses.append(float(loss))
        torch.save({"losses": losses, "params": [p.detach().cpu().clone() for p in params]}, os.path.join(tmp, f"rec{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_recorded_step_equals_single_rank_eager_step(tmp_path):
    """Two ranks, each replaying its three graph segments on its share of the windows with the two gradient
    all-reduces issued between them (the f / g bucket while h's backward pass runs), take the steps one rank takes
    eagerly on all the windows: same losses, same weights on both ranks."""
    import socket
    import torch.multiprocessing as mp
    from muzero_amd.learner import DataParallelLearner
    from tests.test_gpu_nets import _c4_nets
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_recorded_dp_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    nets = [n.cuda() for n in _c4_nets(0)]
    params = [p for n in nets for p in n.parameters()]
    single = DataParallelLearner(*nets, torch.optim.Adam(params, lr=1e-3, capturable=True), 5, (1.0, 0.5, 2.0))
    losses = [single.step_local(b)[0] for b in _batches(3, 24, seed=21)]
    r0 = torch.load(tmp_path / "rec0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "rec1.pt", weights_only=True)
    np.testing.assert_allclose(r0["losses"], losses, rtol=1e-5)
    np.testing.assert_allclose(r1["losses"], losses, rtol=1e-5)
    for a, b, c in zip(r0["params"], r1["params"], params):
        assert torch.equal(a, b)
        # Adam divides by sqrt(v): where a gradient element is small beside its summands, the summation order of the
        # two half-batches shows at the scale of a step (lr = 1e-3); the bound is 1e-2 of one step
        torch.testing.assert_close(a, c.detach().cpu(), rtol=1e-5, atol=1e-5)
