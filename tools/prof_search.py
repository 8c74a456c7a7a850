"""Runs the fused search a few times (for rocprofv3 counter collection); --store-latents: every node's latent kept."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from muzero_amd.engine import BatchedMCTS, PackedNetworks
cfg, h, g, f = bench.build_networks()
G, S = 4096, 50
nets = PackedNetworks(g, f)
mcts = BatchedMCTS(nets, G, S, 1, 0.998, 0.25, keep_latents="--store-latents" in sys.argv)
latent, to_play = bench.synthetic_roots(h, G, 100, torch.device("cuda"))
eta = torch.from_numpy(np.random.default_rng(7).dirichlet(np.full(7, 0.3), size=G).astype(np.float32)).cuda()
rand = torch.from_numpy(bench.child_choice_streams(G, S, 0)).cuda()
for _ in range(3):
    mcts.run(latent, to_play, eta, rand)
torch.cuda.synchronize()
