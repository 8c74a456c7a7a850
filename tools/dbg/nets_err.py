"""Debug aid: prints the largest deviation of every network output from torch fp32 (scale-relative)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.test_gpu_nets import _c4_nets
from muzero_amd.engine import PackedNetworks

_, g, f = _c4_nets()
nets = PackedNetworks(g, f)
for B in (1, 16, 33, 4096):
    torch.manual_seed(B)
    lat = torch.randn(B, 4, 6, 7)
    act = torch.randint(0, 7, (B,))
    with torch.no_grad():
        want_next, want_r = g(lat, act)
        want_p, want_v = f(lat)
    d_lat = lat.reshape(B, -1).cuda().contiguous()
    nxt = torch.zeros(B, 168, device="cuda"); r = torch.zeros(B, device="cuda")
    nets.dynamics(d_lat, B, action=act.int().cuda(), next_latent=nxt, reward=r)
    p = torch.zeros(B, 7, device="cuda"); v = torch.zeros(B, device="cuda")
    nets.prediction(d_lat, B, policy=p, value=v)
    torch.cuda.synchronize()
    def err(a, b):
        a = a.cpu().double().numpy(); b = b.double().numpy()
        return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))
    print(B, "next %.2e reward %.2e policy %.2e value %.2e" % (err(nxt, want_next.reshape(B, -1)), err(r, want_r[:, 0]), err(p, want_p), err(v, want_v[:, 0])), flush=True)
    if B == 1:
        print(nxt[0, :8].cpu().numpy(), want_next.reshape(B, -1)[0, :8].numpy())
