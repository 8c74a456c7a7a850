"""Diagnostic: per-stage shader-clock cycles of the fused search kernel (workgroup 0)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench

NAMES = {0: "staging", 1: "select", 2: "leaf gather", 3: "policy1|merge (all waves)", 4: "E policy rest", 5: "E softmax/priors",
         6: "E child store", 7: "E reward L1 (children)", 8: "E reward rest", 9: "E commit", 10: "E wait for S", 11: "backup",
         12: "result", 16: "S next latent", 17: "S reward head", 18: "S policy1|merge", 19: "S policy rest", 20: "S argmax",
         21: "S value head", 22: "S wait for E"}

# the sequential-program kernel (implicit child latents) stamps its own steps
SEQ_NAMES = {0: "staging", 1: "select + draw", 2: "leaf latent", 3: "merge (leaf)", 16: "child latent",
             18: "policy 1 (leaf + child) | merge (child)", 4: "policy 2 (leaf + child)", 5: "priors", 19: "policy 3 (narrow)",
             6: "rollout latent", 7: "value 1 | reward 1", 8: "value 2", 9: "after the rounds: rewards of all children",
             11: "rollout sums, backup"}


def main():
    from muzero_amd.engine import BatchedMCTS, PackedNetworks
    global NAMES
    cfg, h, g, f = bench.build_networks()
    G, S = int(os.environ.get("MZ_STAGE_GAMES", "4096")), 50  # (256 games: the whole pool fits in L2)
    nets = PackedNetworks(g, f)
    stored = "--store-latents" in sys.argv
    if not stored and os.environ.get("MZ_SEARCH_SEQ", "1") != "0":
        NAMES = SEQ_NAMES
    mcts = BatchedMCTS(nets, G, S, 1, 0.998, 0.25, keep_latents=stored)
    latent, to_play = bench.synthetic_roots(h, G, 100, torch.device("cuda"))
    eta = torch.from_numpy(np.random.default_rng(7).dirichlet(np.full(7, 0.3), size=G).astype(np.float32)).cuda()
    rand = torch.from_numpy(bench.child_choice_streams(G, S, 0)).cuda()
    for _ in range(2):
        mcts.run(latent, to_play, eta, rand)
    mcts.stamps = torch.zeros(96, dtype=torch.int64, device="cuda")
    mcts.run(latent, to_play, eta, rand)
    torch.cuda.synchronize()
    st = mcts.stamps.cpu().numpy()
    tot = st[:32].sum()
    for k in sorted(NAMES):
        v = st[k]
        print(f"{NAMES[k]:28s} {v:12d} cycles  {100*v/tot:5.1f}%   {v/(S+1):9.0f} /round")
    print("total (thread 0)", tot)
    tl = st[32:].reshape(8, 8)
    if tl.any():  # diagnostic build (-DMZ_DENSE_TIMELINE): the last matrix-layer call of each wave
        print("wave: entry->issue, ->first fetch issued, ->first batch consumed, ->last batch consumed, ->epilogue done, ->after sync")
        for w in range(8):
            t = tl[w]
            print(w, [int(t[i + 1] - t[i]) for i in range(6)], "(slots 0..6)")


main()
