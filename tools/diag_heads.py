"""Timing aid: the head MLPs of the shipped tower shape on their own (one graph of many launches each), to tell the
matrix chain of a head from its post-processing.  python tools/diag_heads.py [rows]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muzero_amd import _hip  # noqa: E402
from muzero_amd._hip import ptr  # noqa: E402
from muzero_amd.towers import TowerNetworks  # noqa: E402
from tools.bench_towers import build_nets  # noqa: E402


def timed(fn, n=200):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    A = 5
    g, f = build_nets((256, 6, 6), A)
    nets = TowerNetworks(g, f)
    lib = _hip.lib()
    dt = nets.bottleneck_f
    t = torch.randn(B, dt, device="cuda")
    pol = torch.zeros(B, A, device="cuda")
    am = torch.zeros(B, dtype=torch.int32, device="cuda")
    val = torch.zeros(B, device="cuda")
    st = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731
    cases = {
        "policy head, softmax x1 + argmax (as the search calls it)": lambda: lib.mz_mlp_forward_post(
            C.byref(nets.policy_t), ptr(nets.blob), ptr(t), dt, None, ptr(pol), A, B, 1, ptr(am), st()),
        "policy head, argmax only": lambda: lib.mz_mlp_forward_post(
            C.byref(nets.policy_t), ptr(nets.blob), ptr(t), dt, None, None, A, B, 0, ptr(am), st()),
        "policy head, plain": lambda: lib.mz_mlp_forward(
            C.byref(nets.policy_t), ptr(nets.blob), ptr(t), dt, None, ptr(pol), A, B, st()),
        "value head, plain": lambda: lib.mz_mlp_forward(
            C.byref(nets.value_t), ptr(nets.blob), ptr(t), dt, None, ptr(val), 1, B, st()),
    }
    for name, fn in cases.items():
        print(f"{B:6d} rows  {timed(fn):7.2f} us  {name}")
    for L in nets.policy_t.layer[:nets.policy_t.n_layers]:
        print("   policy layer", L.K, "->", L.N)


if __name__ == "__main__":
    main()
