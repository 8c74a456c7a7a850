"""g's rollout-step tower call in the search form (bottleneck output only), a dozen launches: the
program rocprofv3 --pmc passes are run on (FETCH_SIZE and WRITE_SIZE in separate passes).

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- python3 tools/pmc_tower_step.py [trees]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from muzero_amd.towers import TowerNetworks  # noqa: E402
from tools.bench_towers import build_nets  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
A = 5
g, f = build_nets((256, 6, 6), A)
nets = TowerNetworks(g, f)
D, dt = nets.D, nets.bottleneck_g
base = torch.randn(G, D, device="cuda")
t_out = torch.empty(G, dt, device="cuda")
sel = torch.zeros(G, dtype=torch.int32, device="cuda")
for _ in range(12):
    nets._tower_call(nets.tower_g, base, D, None, nets.table_t, sel, None, 0, None, G, 1, t_out=t_out)
torch.cuda.synchronize()
print("algorithmic bytes per launch", 4 * G * (D + dt))
