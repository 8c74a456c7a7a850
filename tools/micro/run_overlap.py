import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "..", "_exp", "libov.so"))
lib.ov_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
sink = torch.zeros(256 * 512, device="cuda"); out = torch.zeros(8, dtype=torch.int64, device="cuda")
iters = 2000
for mode, name in ((1, "mfma only (16/iter on waves 0-3)"), (2, "valu only (128 fma/iter on waves 4-7)"), (3, "both")):
    lib.ov_run(mode, 100, sink.data_ptr(), out.data_ptr())
    lib.ov_run(mode, iters, sink.data_ptr(), out.data_ptr())
    o = out.cpu().numpy() / iters
    print(f"{name:40s} cycles/iter per wave:", " ".join(f"{v:7.1f}" for v in o))
