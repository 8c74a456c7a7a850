"""Runs the wide dense layer alone a few times (for rocprofv3 kernel traces and counter passes).

    python tools/prof_wide.py [M] [K] [N] [repeats]        default 256 9216 9216 5
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from muzero_amd import _hip  # noqa: E402
from muzero_amd._hip import ptr  # noqa: E402

M, K, N, REP = ([int(v) for v in sys.argv[1:]] + [256, 9216, 9216, 5][len(sys.argv) - 1:])[:4]
lib = _hip.lib()
torch.manual_seed(0)
w = torch.randn(N, K, device="cuda") / K ** 0.5
x = torch.randn(M, K, device="cuda")
d = _hip.Dense(ptr(w), None, K, N, _hip.ACT["none"], 0)
pl = _hip.PackedLayer()
n = lib.mz_wide_plan(C.byref(d), 0, C.byref(pl))
blob = torch.zeros(n, device="cuda")
st = torch.cuda.current_stream().cuda_stream
_hip.check(lib.mz_wide_pack(C.byref(d), C.byref(pl), ptr(blob), st), "pack")
need = lib.mz_wide_workspace(C.byref(pl), M)
ws = torch.empty(max(need, 1), device="cuda")
y = torch.empty(M, N, device="cuda")
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(REP + 1):
    if i == 1:
        ev0.record()
    _hip.check(lib.mz_wide_forward(C.byref(pl), ptr(blob), ptr(x), K, None, ptr(y), N, M, ptr(ws), need, st), "forward")
ev1.record()
torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / REP
ref = x[:8].double() @ w.double().t()
err = float((y[:8].double() - ref).abs().max())
print(f"M={M} K={K} N={N}: {ms * 1e3:.1f} us per call (finish kernel included), "
      f"{2.0 * M * K * N / ms / 1e9:.1f} TFLOP/s, workspace {need * 4 / 1e6:.1f} MB, max abs err {err:.2e}")
