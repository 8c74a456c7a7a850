"""Times mz_merge_conv_forward (fc_merge + first convolution of g's tower) and mz_tower_tail_forward alone at the
shipped tower shape: the rollout-step form (one action per row) and the expansion form (all A actions per row).
   python tools/bench_merge_conv.py [games]       MZ_MC_SKIP=<mask> leaves parts of the kernel out (timing aid)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from muzero_amd import _hip  # noqa: E402
from muzero_amd._hip import ptr  # noqa: E402
from tools.bench_towers import build_nets  # noqa: E402
from muzero_amd.towers import TowerNetworks  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
A = 5
g, f = build_nets((256, 6, 6), A)
nets = TowerNetworks(g, f)
assert nets.fused_g
lib = _hip.lib()
st = torch.cuda.current_stream().cuda_stream
c, hgt, wid = nets.shape
hw = hgt * wid
t = torch.randn(G, 108, device="cuda")
act = torch.randint(0, A, (G,), dtype=torch.int32, device="cuda")
Y = torch.empty(G * A, hw, 32, device="cuda")
t_out = torch.empty(G * A, 108, device="cuda")
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(name, call, flop):
    call()
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(20):
        call()
    ev1.record()
    torch.cuda.synchronize()
    us = ev0.elapsed_time(ev1) / 20 * 1e3
    print(f"{name:50s} {us:8.1f} us   {flop / us / 1e6:7.1f} TFLOP/s")


def merge_conv(action):
    def call():
        if nets.split_merge_conv:
            _hip.check(lib.mz_merge_conv_forward_split(C.byref(nets.merge_t_perm), ptr(nets.split3["merge_t_perm"]), C.byref(nets.conv_g),
                                                       ptr(nets.split3["conv_g"]), ptr(t), 108, None, ptr(nets.table_t_perm), ptr(action),
                                                       A, G, c, hw, ptr(Y), st), "mz_merge_conv_forward_split")
        else:
            _hip.check(lib.mz_merge_conv_forward(C.byref(nets.merge_t_perm), C.byref(nets.conv_g), ptr(nets.blob), ptr(t), 108, None,
                                                 ptr(nets.table_t_perm), ptr(action), A, G, c, hw, ptr(Y), st), "mz_merge_conv_forward")
    return call


def tail(B, group):
    def call():
        _hip.check(lib.mz_tower_tail_forward(C.byref(nets.tower_g), ptr(Y), B, group, ptr(t_out), 108, None, None, st), "tail")
    return call


merge_flop = 2.0 * G * 108 * c * hw
conv_flop = 2.0 * G * c * hw * 32
timed("merge + conv, one action per row", merge_conv(act), merge_flop + conv_flop)
timed("merge + conv, all actions (expansion)", merge_conv(None), merge_flop + A * conv_flop)
timed("tower tail, group 1", tail(G, 1), 1.0)
timed("tower tail, group A", tail(G * A, A), 1.0)
