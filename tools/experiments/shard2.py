import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import dhg_amd
from dhg_amd import spec
B, L, Lt = 64, 488, 30
sd = {k: torch.from_numpy(v) for k, v in spec.synthetic_state_dict(2).items()}
m = dhg_amd.DiffusionModel(2, precision="bf16", max_B=B, max_L=L, max_Lt=Lt).eval(); m.load_state_dict(sd)
inp = spec.synthetic_inputs(B, L, Lt, seed=1)
tx, sv = torch.from_numpy(inp["text"]).cuda(), torch.from_numpy(inp["style"]).cuda()
bad = 0
for T in (2, 3, 4, 5, 6, 7, 8, 9):
    nz = torch.from_numpy(spec.synthetic_inputs(B, L, Lt, seed=1, T=T)["noise"]).cuda()
    f = dhg_amd.sample(m, tx, sv, L=L, T=T, noise=nz).cpu()
    sh = dhg_amd.sample(m, tx[40:48].contiguous(), sv[40:48].contiguous(), L=L, T=T, noise=nz[:, 40:48].contiguous()).cpu()
    d = (sh - f[40:48]).abs()
    if d.max() > 0:
        bad += 1
        print("T", T, "max diff", d.max().item(), "samples", sorted(set(torch.nonzero(d)[:, 0].tolist())), flush=True)
print(os.environ.get("TAG"), "mismatching T values:", bad, "of 8")
