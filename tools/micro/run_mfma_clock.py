"""Sustained shader clock and fp32 matrix throughput of the whole chip under a pure MFMA stream
(256 workgroups x 8 waves, independent accumulators), for pricing rooflines against what the
part sustains rather than its boost clock."""
import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "..", "_exp", "librate.so"))
lib.rate_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
sink = torch.zeros(256 * 1024, device="cuda"); out = torch.zeros(16, dtype=torch.int64, device="cuda")
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for iters in (20000, 200000, 1000000):
    lib.rate_run(16, 8, 1000, sink.data_ptr(), out.data_ptr())
    ev0.record()
    lib.rate_run(16, 8, iters, sink.data_ptr(), out.data_ptr())
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1)
    cyc = float(out.cpu().numpy()[:8].max())
    flop = 256 * 8 * iters * 16 * 2048.0
    print(f"iters {iters}: {ms:.2f} ms, {cyc / ms / 1e6:.3f} GHz by the shader clock counter, {flop / ms / 1e9:.1f} TFLOP/s")
