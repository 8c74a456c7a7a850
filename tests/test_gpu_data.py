"""GPU parity of the data-side kernels: history ring/tensor and z-targets bit-exact against
the reference-generated vectors and the CPU oracle; the 1x1-conv representation network and the
action pick against torch."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import lib as oracle_lib
from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu
DATA = np.load(os.path.join(GOLDEN, "data_path.npz"))


def test_history_kernel_matches_reference_vectors():
    """Two games run the recorded sequences with a one-step lag between them (different ring
    phases in one launch); every step's history tensor must equal the reference's bit for bit."""
    from muzero_amd.selfplay import DeviceHistory
    hi0 = 0
    for K in (1, 3, 4):
        frames, actions = DATA[f"histseq_K{K}_frames"][:, 0], DATA[f"histseq_K{K}_actions"]
        T, shape = frames.shape[0], frames.shape[1:]
        hist = DeviceHistory(2, K, tuple(shape))
        for t in range(T + 1):
            t0, t1 = min(t, T - 1), max(t - 1, 0)  # game 1 lags one step
            fr = torch.from_numpy(np.stack([frames[t0], frames[t1]])).cuda().contiguous()
            ac = torch.from_numpy(np.array([actions[t0], actions[t1]], np.int32)).cuda()
            st = torch.tensor([t0 if t < T else T, t1 if t > 0 else 0], dtype=torch.int32, device="cuda")
            if t == T:
                st[0] = T - 1  # game 0 idles: re-pushing the same step is idempotent
            out = hist.push(fr, st, ac).cpu().numpy()
            if t < T:
                assert np.array_equal(out[0:1], DATA[f"hist{hi0 + t}_out"]), (K, t)
            if t > 0:
                assert np.array_equal(out[1:2], DATA[f"hist{hi0 + t - 1}_out"]), (K, t)
        hi0 += T


def test_z_target_kernel_bit_exact():
    from muzero_amd.training import compute_z_targets_batch
    for i in range(int(DATA["n_z"])):
        w, gamma = int(DATA[f"z{i}_w"]), float(DATA[f"z{i}_gamma"])
        z = compute_z_targets_batch(DATA[f"z{i}_R"][None], DATA[f"z{i}_V"][None], gamma)
        assert np.array_equal(z[0], DATA[f"z{i}_z"])
    # a large random batch against the CPU oracle
    rng = np.random.default_rng(0)
    B, w, gamma = 4096, 5, 0.997
    R, V = rng.standard_normal((B, w + 1)), rng.standard_normal((B, w + 1))
    z = compute_z_targets_batch(R, V, gamma)
    want = np.zeros_like(z)
    dp = C.POINTER(C.c_double)
    for b in range(B):
        oracle_lib().mzo_z_targets(np.ascontiguousarray(R[b]).ctypes.data_as(dp), np.ascontiguousarray(V[b]).ctypes.data_as(dp),
                                   C.c_int32(w), C.c_double(gamma), want[b].ctypes.data_as(dp))
    assert np.array_equal(z, want)


def test_pointwise_representation_matches_torch_and_reference_vectors():
    from muzero_amd.selfplay import RepresentationRunner
    from tests.test_gpu_nets import _c4_nets
    h, _, _ = _c4_nets(0)
    run = RepresentationRunner(h)
    assert run.pointwise
    z = np.load(os.path.join(GOLDEN, "c4_mlp.npz"))
    out = run(torch.from_numpy(z["h_in"]).cuda()).cpu().numpy()
    np.testing.assert_allclose(out, z["h_out"], rtol=1e-5, atol=1e-6)
    x = torch.randn(300, 3, 6, 7)
    with torch.no_grad():
        np.testing.assert_allclose(run(x.cuda()).cpu().numpy(), h(x).numpy(), rtol=1e-5, atol=1e-6)


def test_pick_action_first_argmax_and_exploration():
    from muzero_amd.selfplay import pick_action
    G, A = 1000, 7
    rng = np.random.default_rng(2)
    pol = rng.random((G, A)).astype(np.float32)
    pol[::3, 2] = pol[::3, 5] = 2.0  # exact ties: the first maximum must win
    explore = (rng.random(G) < 0.3).astype(np.int8)
    rnd = rng.integers(0, A, size=G).astype(np.int32)
    out = torch.zeros(G, dtype=torch.int32, device="cuda")
    pick_action(torch.from_numpy(pol).cuda(), torch.from_numpy(explore).cuda(), torch.from_numpy(rnd).cuda(), out)
    want = np.where(explore == 1, rnd, torch.from_numpy(pol).argmax(dim=1).numpy())
    assert np.array_equal(out.cpu().numpy(), want)
    assert (want[::3][explore[::3] == 0] == 2).all()
