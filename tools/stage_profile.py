"""Diagnostic: per-stage shader-clock cycles of the fused search kernel (workgroup 0)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench

NAMES = ["staging", "select", "leaf gather", "E policy1|merge", "E policy rest", "E softmax/priors", "E child store",
         "E reward L1 (children)", "E reward rest", "commit", "S next latent", "S reward head", "S policy1|merge",
         "S policy rest", "S argmax", "S value head", "backup", "result"]

def main():
    from muzero_amd.engine import BatchedMCTS, PackedNetworks
    cfg, h, g, f = bench.build_networks()
    G, S = 4096, 50
    nets = PackedNetworks(g, f)
    mcts = BatchedMCTS(nets, G, S, 1, 0.998, 0.25)
    latent, to_play = bench.synthetic_roots(h, G, 100, torch.device("cuda"))
    eta = torch.from_numpy(np.random.default_rng(7).dirichlet(np.full(7, 0.3), size=G).astype(np.float32)).cuda()
    rand = torch.from_numpy(bench.child_choice_streams(G, S, 0)).cuda()
    for _ in range(2):
        mcts.run(latent, to_play, eta, rand)
    mcts.stamps = torch.zeros(24, dtype=torch.int64, device="cuda")
    mcts.run(latent, to_play, eta, rand)
    torch.cuda.synchronize()
    st = mcts.stamps.cpu().numpy()
    tot = st.sum()
    for n, v in zip(NAMES, st):
        print(f"{n:28s} {v:12d} cycles  {100*v/tot:5.1f}%   {v/(S+1):9.0f} /round")
    print("total", tot)
    import ctypes
    from muzero_amd import _hip
    lib = ctypes.CDLL(_hip.LIB_PATH)
    if hasattr(lib, "mz_debug_read"):
        buf = (ctypes.c_ulonglong * 64)()
        lib.mz_debug_read(buf, 0)
        names = ["fetch(wb)", "compute(wa)", "compute(wb)", "whole call"]
        for base, tag in ((0, "MT=1 NTW=2"), (8, "MT=1 NTW=1"), (16, "MT>1 NTW=2"), (24, "MT>1 NTW=1")):
            for k, n in enumerate(names):
                c, cnt = buf[base + 2 * k], buf[base + 2 * k + 1]
                if cnt:
                    print(f"{tag:12s} {n:12s} cycles {c:10d} calls {cnt:7d} avg {c / cnt:9.0f}")

main()
