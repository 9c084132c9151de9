import sys, time
sys.path.insert(0, "/root/repo")
import torch, dhg_amd
from dhg_amd import spec
B, L, Lt, T = 32, 1000, 62, 1000
m = dhg_amd.DiffusionModel(2, precision="bf16", max_B=B, max_L=L, max_Lt=Lt).eval()
m.load_state_dict({k: torch.from_numpy(v) for k, v in spec.synthetic_state_dict(2).items()})
inp = spec.synthetic_inputs(B, L, Lt, seed=5, T=0)
tx, sv = torch.from_numpy(inp["text"]).cuda(), torch.from_numpy(inp["style"]).cuda()
for k in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    out = dhg_amd.sample(m, tx, sv, L=L, T=T, seed=k)
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"B={B} L={L} T={T}: {dt*1e3:.1f} ms, finite={bool(torch.isfinite(out).all())}, max|x|={out[..., :2].abs().max().item():.3g}, {B*L/dt:.4g} stroke-points/s", flush=True)
