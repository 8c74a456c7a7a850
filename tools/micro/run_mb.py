"""Runs the dense-routine microbenchmark (tools/micro/mb_dense.hip built into _exp/libmb.so)."""
import ctypes, os, sys, torch
lib = ctypes.CDLL(os.environ.get("MB_LIB", os.path.join(os.path.dirname(__file__), "..", "..", "_exp", "libmb.so")))
lib.mb_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
blob = torch.randn(200000, device="cuda") * 0.05
out = torch.zeros(256, dtype=torch.int64, device="cuda")
names = ["R1 children MT7 K176 N128", "P1|M MT1 K176 N432", "P2 MT1 K256 N256", "V2 MT1 K256 N128", "R1 MT1 K176 N128"]
pipe = [7 * 8 * 44 * 32 / 4, 27 * 44 * 32 / 4, 16 * 64 * 32 / 4, 8 * 64 * 32 / 4, 8 * 44 * 32 / 4]
names += [None] * 5 + ["E alone: R1c MT4 x2 (4 waves)", "S alone: P1|M, P2, P2 (4 waves)", "E and S together", "R1c MT7 on 4 waves"]
pipe += [0] * 5 + [2 * 4 * 8 * 44 * 32 / 4, (27 * 44 + 2 * 16 * 64) * 32 / 4, (2 * 4 * 8 * 44 + 27 * 44 + 2 * 16 * 64) * 32 / 4, 7 * 8 * 44 * 32 / 4]
for blocks in (256,):
    for v, n in enumerate(names):
        if n is None:
            continue
        lib.mb_run(v, 20, blob.data_ptr(), out.data_ptr(), blocks)
        rc = lib.mb_run(v, 200, blob.data_ptr(), out.data_ptr(), blocks)
        torch.cuda.synchronize()
        c = int(out[0])
        print(f"blocks={blocks} {n:28s} rc={rc} cycles/call {c:7d}  pipe-bound {pipe[v]:7.0f}  eff {pipe[v]/c:5.2f}")
        if os.environ.get("MB_TL") and v in (2, 13):
            tl = out[8:136].cpu().numpy().reshape(8, 16)
            t0 = tl[:, 0].min()
            print("   wave: entry  fetch0  fetched  comp(wa)0  comp(wb)0  loopend  pre-epi  ret  barrier  (cycles since first entry; NTW=1 pass overwrites NTW=2)")
            for w in range(8):
                print("   %d: " % w + " ".join("%6d" % (tl[w, k] - t0) for k in (0, 1, 2, 3, 4, 5, 6, 7, 8)))
