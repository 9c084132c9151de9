import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import dhg_amd
from dhg_amd import spec
B, L, Lt = 64, 488, 30
sd = {k: torch.from_numpy(v) for k, v in spec.synthetic_state_dict(2).items()}
def model(env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    m = dhg_amd.DiffusionModel(2, precision="bf16", max_B=B, max_L=L, max_Lt=Lt).eval(); m.load_state_dict(sd)
    return m, old
inp = spec.synthetic_inputs(B, L, Lt, seed=1)
tx, sv = torch.from_numpy(inp["text"]).cuda(), torch.from_numpy(inp["style"]).cuda()
strokes = torch.from_numpy(inp["strokes"]).cuda()
sg = torch.full((B, 1), 0.7).cuda()
m, _ = model({})
e_full, p_full, _ = m(strokes, tx, sg, sv)
for env in ({}, {"DHW_ENC_BM256": "32"}, {"DHW_ENC_BM256": "32", "DHW_ENC_BM192": "64"}, {"DHW_ENC_BM256": "32", "DHW_ENC_BM192": "64", "DHW_CONV_BM": "64"}):
    os.environ.update(env)
    e_sub, p_sub, _ = m(strokes[40:48].contiguous(), tx[40:48].contiguous(), sg[40:48].contiguous(), sv[40:48].contiguous())
    print(env, "max diff eps", (e_sub - e_full[40:48]).abs().max().item(), "pen", (p_sub - p_full[40:48]).abs().max().item(), flush=True)
    for k in env: os.environ.pop(k, None)
for name in ("enc1", "enc2", "enc3", "enc4", "enc5", "att_layers.0", "att_layers.1", "dec3", "dec2", "dec1"):
    pass
full = dhg_amd.sample(m, tx, sv, L=L, seed=7).cpu()
again = dhg_amd.sample(m, tx, sv, L=L, seed=7).cpu()
print("again equal", torch.equal(full, again))
for T in (1, 2, 5, 60):
    f = dhg_amd.sample(m, tx, sv, L=L, T=T, seed=7).cpu()
    sh = dhg_amd.sample(m, tx[40:48].contiguous(), sv[40:48].contiguous(), L=L, T=T, seed=7, first_sample=40).cpu()
    d = (sh - f[40:48]).abs()
    print("T", T, "max diff", d.max().item(), "n diff", int((d > 0).sum()), "of", d.numel(), "max|x|", f.abs().max().item(), flush=True)
    nz = torch.from_numpy(spec.synthetic_inputs(B, L, Lt, seed=1, T=T)["noise"]).cuda()
    f = dhg_amd.sample(m, tx, sv, L=L, T=T, noise=nz).cpu()
    sh = dhg_amd.sample(m, tx[40:48].contiguous(), sv[40:48].contiguous(), L=L, T=T, noise=nz[:, 40:48].contiguous()).cpu()
    d = (sh - f[40:48]).abs()
    print("  external noise: max diff", d.max().item(), "n diff", int((d > 0).sum()), flush=True)
