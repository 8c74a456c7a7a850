"""Helpers shared by the GPU parity tests."""
import numpy as np
import torch


def random_streams(G, A, S, depth, seed=0, value_scale=0.1, tie_fraction=0.1):
    """Synthetic recorded network outputs: normalised U(0,1) priors (a fraction of trees get
    exactly equal priors to exercise first-index tie-breaking), N(0, value_scale) rewards
    and values, uniform child choices, Dirichlet(0.3)-like eta."""
    rng = np.random.default_rng(seed)
    p = rng.random((G, S + 1, A), dtype=np.float32) + 1e-3
    p /= p.sum(-1, keepdims=True)
    ties = rng.random(G) < tie_fraction
    p[ties] = np.float32(1.0 / A)
    rew = (rng.standard_normal((G, S + 1, A)) * value_scale).astype(np.float32)
    sim = (rng.standard_normal((G, S, depth + 1)) * value_scale).astype(np.float32)
    big = rng.random(G) < 0.25
    sim[big] *= 40.0  # value-dominated trees
    rand = rng.integers(0, A, size=(G, S + 1)).astype(np.int32)
    eta = rng.gamma(0.3, size=(G, A)).astype(np.float32) + 1e-6
    eta /= eta.sum(-1, keepdims=True)
    to_play = rng.choice(np.array([1, -1, 0], np.int8), size=G)
    return dict(prior=p, reward=rew, sim=sim, rand=rand, eta=eta.astype(np.float32), to_play=to_play)


def dev(a, device="cuda"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def streams_to_device(st):
    """[G, ...]-major numpy streams -> step-major device tensors used by engine.run_recorded."""
    return dict(
        prior=dev(st["prior"].transpose(1, 0, 2)), reward=dev(st["reward"].transpose(1, 0, 2)),
        rand=dev(st["rand"].T), sim=dev(st["sim"].transpose(1, 0, 2)), eta=dev(st["eta"]),
        to_play=dev(st["to_play"]))
