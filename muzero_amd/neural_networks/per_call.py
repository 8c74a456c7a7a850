"""Per-call batch statistics for a batch of independent network calls.

The reference evaluates h, g and f one game (or one training window) at a time and never
leaves train mode, so every BatchNorm2d normalises with the statistics of that single call
(src/main.py:39 only disables gradients; src/training.py:244-305 unrolls one window per
call).  A batched evaluation reproduces that by giving every sample its own statistics.
"""
from __future__ import annotations

import torch


def _batchnorm_per_sample(bn: torch.nn.BatchNorm2d, x: torch.Tensor) -> torch.Tensor:
    """BatchNorm2d in train mode applied to every sample of the batch on its own: folding the
    batch into the channel axis gives each (sample, channel) plane its own mean and biased
    variance, which is what B separate calls with batch size 1 compute."""
    G, Cn, H, W = x.shape
    w = bn.weight.repeat(G) if bn.weight is not None else None
    b = bn.bias.repeat(G) if bn.bias is not None else None
    y = torch.nn.functional.batch_norm(x.reshape(1, G * Cn, H, W), None, None, w, b, True, 0.0, bn.eps)
    return y.reshape(G, Cn, H, W)


class per_call_statistics:
    """Context manager: inside it every BatchNorm2d of `module` normalises each sample with that
    sample's own statistics.  The reference evaluates h once per game and move with batch size 1
    and never leaves train mode (src/training_data_generator.py:114-118, main.py:39), so a batch
    of G games must not share statistics.  Running statistics are left untouched."""

    def __init__(self, module: torch.nn.Module):
        self.bns = [m for m in module.modules() if isinstance(m, torch.nn.BatchNorm2d)]

    def __enter__(self):
        for bn in self.bns:
            bn.forward = (lambda x, _bn=bn: _batchnorm_per_sample(_bn, x))
        return self

    def __exit__(self, *exc):
        for bn in self.bns:
            del bn.forward
        return False
