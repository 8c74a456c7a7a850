import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "..", "_exp", "librate.so"))
lib.rate_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
sink = torch.zeros(256 * 1024, device="cuda"); out = torch.zeros(16, dtype=torch.int64, device="cuda")
iters = 4000
for kind, nacc, name, flop in ((16, 16, "16x16x4  16 accumulators", 2048), (164, 4, "16x16x4   4 accumulators", 2048),
                               (32, 4, "32x32x2   4 accumulators", 4096), (322, 2, "32x32x2   2 accumulators", 4096)):
    for waves in (4, 8, 16):
        lib.rate_run(kind, waves, 100, sink.data_ptr(), out.data_ptr())
        assert lib.rate_run(kind, waves, iters, sink.data_ptr(), out.data_ptr()) == 0
        cyc = out.cpu().numpy()[:waves].max() / (iters * nacc)
        per_simd = waves / 4
        print(f"{name}  {waves:2d} waves/CU: {cyc:6.1f} cycles per instruction per wave -> "
              f"{flop * per_simd / cyc:6.1f} flop/cycle/SIMD (peak 64)")
