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
    for p, q in zip(pa, pb):
        torch.testing.assert_close(p, q, rtol=1e-6, atol=1e-8)
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
