"""One two-plane fp16 convolution kernel alone: mz_conv3_same_forward (default) or mz_conv3_forward
(--gather) on G images of H x W pixels, C -> N channels, 3 x 3, timed with HIP events.

    python tools/bench_conv3_kernel.py [--games 256] [--hw 48] [--cin 128] [--cout 128] [--gather] [--raw]

Prints one JSON line: microseconds per launch and the executed fp16 matrix rate (three products per
float32 product) against the 2.5 PFLOP/s dense fp16 peak.  MUZERO_HIP_LIB selects another build of
the library (diagnostic variants)."""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F16_PEAK_TFLOPS = 2500.0


def main():
    from muzero_amd import _hip
    from muzero_amd._hip import ptr
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=256)
    ap.add_argument("--hw", type=int, default=48)
    ap.add_argument("--width", type=int, default=0, help="image width when it differs from the height (--hw)")
    ap.add_argument("--cin", type=int, default=128)
    ap.add_argument("--cout", type=int, default=128)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--gather", action="store_true")
    ap.add_argument("--raw", action="store_true", help="float32 input with BatchNorm scale / shift applied while staging")
    a = ap.parse_args()
    lib = _hip.lib()
    G, H, W, C, N, k = a.games, a.hw, a.width or a.hw, a.cin, a.cout, 3
    rows, taps = G * H * W, k * k
    gen = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(rows * C, device="cuda", generator=gen)
    w = torch.randn(N * taps * C, device="cuda", generator=gen) * 0.05
    x3 = torch.empty((2, rows * C), dtype=torch.int16, device="cuda")
    w3 = torch.empty((2, N * taps * C), dtype=torch.int16, device="cuda")
    _hip.check(lib.mz_split3(ptr(x), x.numel(), ptr(x3), x3.shape[1], None), "mz_split3")
    _hip.check(lib.mz_split3(ptr(w), w.numel(), ptr(w3), w3.shape[1], None), "mz_split3")
    y = torch.empty(rows * N, device="cuda")
    scale = torch.rand((G, C), device="cuda", generator=gen) + 0.5
    shift = torch.randn((G, C), device="cuda", generator=gen)
    idx = None
    if a.gather:
        idx = torch.empty((taps, rows), dtype=torch.int32, device="cuda")
        _hip.check(lib.mz_conv_index(ptr(idx), G, H, W, k, 1, 1, H, W, None), "mz_conv_index")

    def call():
        if a.gather:
            _hip.check(lib.mz_conv3_forward(ptr(x3), x3.shape[1], C, ptr(idx), taps, ptr(w3), w3.shape[1], None, 0, ptr(y), None, 0,
                                            rows, N, None, None), "mz_conv3_forward")
        elif a.raw:
            _hip.check(lib.mz_conv3_same_bn_forward(ptr(x), ptr(scale), ptr(shift), 1, C, G, H, W, k, ptr(w3), w3.shape[1], None, 0,
                                                    ptr(y), None, 0, N, None, None), "mz_conv3_same_bn_forward")
        else:
            _hip.check(lib.mz_conv3_same_forward(ptr(x3), x3.shape[1], C, G, H, W, k, ptr(w3), w3.shape[1], None, 0, ptr(y), None, 0,
                                                 N, None, None), "mz_conv3_same_forward")

    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.iters
    if os.environ.get("MZ_CONV3_STAMPS"):  # stamped diagnostic build: cycles per part, mean over waves
        st = y.view(rows, N)[::256, :64].reshape(-1, 8, 8)
        print("stamps (prologue, loop, epilogue cycles; in the loop: stores + requests, barrier, the three terms): ",
              [round(v) for v in st.mean(dim=(0, 1)).tolist()], file=sys.stderr)
    flop = 2.0 * rows * N * taps * C
    print(json.dumps({"kernel": "gather" if a.gather else ("same+bn" if a.raw else "same"), "games": G, "hw": H, "width": W, "cin": C, "cout": N,
                      "us_per_launch": us, "fp32_equivalent_tflops": flop / us / 1e6, "f16_executed_tflops": 3 * flop / us / 1e6,
                      "f16_frac_of_peak": 3 * flop / us / 1e6 / F16_PEAK_TFLOPS, "lib": os.environ.get("MUZERO_HIP_LIB", "default")}))


if __name__ == "__main__":
    main()
