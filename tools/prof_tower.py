"""Times mz_tower_forward alone on the shipped tower shape and on reduced variants (which part
of the kernel the time goes to).   python tools/prof_tower.py [games]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from muzero_amd import _hip  # noqa: E402
from muzero_amd._hip import ptr  # noqa: E402
from tools.bench_towers import build_nets  # noqa: E402
from muzero_amd.towers import TowerNetworks  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 256
A = 5
g, f = build_nets((256, 6, 6), A)
nets = TowerNetworks(g, f)
D = nets.D
lib = _hip.lib()
st = torch.cuda.current_stream().cuda_stream
x = torch.randn(G, D, device="cuda")
y = torch.empty(G * A, D, device="cuda")
gidx = (torch.arange(G * A, dtype=torch.int32, device="cuda") // A).contiguous()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


t_in = torch.randn(G, 108, device="cuda")
t_out = torch.empty(G * A, 108, device="cuda")


def timed(name, tower, B, group, x_rows=None, row_add=None, lean=False, src=None):
    xs = x if src is None else src

    def call():
        _hip.check(lib.mz_tower_forward(C.byref(tower), ptr(xs), xs.shape[1], ptr(x_rows), ptr(row_add), D, None,
                                        None if lean else ptr(y), D, None, ptr(t_out) if lean else None, 108, B, group,
                                        st), name)
    call()
    ev0.record()
    for _ in range(10):
        call()
    ev1.record()
    torch.cuda.synchronize()
    print(f"{name:55s} {ev0.elapsed_time(ev1) / 10 * 1e3:8.1f} us")


timed("f tower, group 1", nets.tower_f, G, 1)
timed("g tower, group 1, merge table", nets.tower_g, G, 1, row_add=nets.table)
timed("g tower, group A (expansion), merge table", nets.tower_g, G * A, A, x_rows=gidx, row_add=nets.table)
# the forms the lock-step search launches (last block's output only, no res_final)
timed("g tower, group 1, bottleneck output only", nets.tower_g, G, 1, row_add=nets.table, lean=True)
timed("g tower, group A, bottleneck output only", nets.tower_g, G * A, A, x_rows=gidx, row_add=nets.table, lean=True)
if nets.tower_f_t is not None:
    timed("f tower on the bottleneck (3 -> 3 block), group 1", nets.tower_f_t, G, 1, lean=True, src=t_in)
copy = _hip.Tower.from_buffer_copy(nets.tower_f)
copy.n_blocks = 0
copy.final_w = copy.final_b = None
timed("no blocks, no res_final (copy only), group 1", copy, G, 1)
timed("no blocks, no res_final (copy only), group A", copy, G * A, A, x_rows=gidx)

if hasattr(lib, "mz_debug_tower_stamps"):  # diagnostics build only
    import numpy as np
    names = ["start", "stage: begin", "stage: loaded", "conv1 done", "before bn1", "after bn1/skip bn", "conv2 done",
             "after bn2", "before res_final", "end"]
    for label, tower, B, group, rows, add, xs in (
            ("g step, bottleneck output only", nets.tower_g, G, 1, None, nets.table, x),
            ("g expansion, bottleneck output only", nets.tower_g, G * A, A, gidx, nets.table, x),
            ("f on the bottleneck", nets.tower_f_t, G, 1, None, None, t_in)):
        _hip.check(lib.mz_tower_forward(C.byref(tower), ptr(xs), xs.shape[1], ptr(rows), ptr(add), D, None, None, D, None,
                                        ptr(t_out), 108, B, group, st), label)
        torch.cuda.synchronize()
        h = np.zeros(16, np.uint64)
        lib.mz_debug_tower_stamps(h.ctypes.data_as(C.c_void_p))
        t = h[:10].astype(np.int64)
        print(label, "(shader-clock cycles since start, workgroup 0; staging stamps are of the last pass)")
        for i in range(1, 10):
            print(f"   {names[i]:22s} {int(t[i] - t[0]):9d}")
