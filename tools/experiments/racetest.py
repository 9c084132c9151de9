import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import dhg_amd
from dhg_amd import spec
B, L, Lt = int(os.environ.get("RB", 8)), 488, 30
sd = {k: torch.from_numpy(v) for k, v in spec.synthetic_state_dict(2).items()}
m = dhg_amd.DiffusionModel(2, precision="bf16", max_B=B, max_L=L, max_Lt=Lt).eval(); m.load_state_dict(sd)
inp = spec.synthetic_inputs(B, L, Lt, seed=1)
tx, sv = torch.from_numpy(inp["text"]).cuda(), torch.from_numpy(inp["style"]).cuda()
strokes = torch.from_numpy(inp["strokes"]).cuda()
sg = torch.full((B, 1), 0.7).cuda()
ref = None
bad = 0
for i in range(int(os.environ.get("RN", 40))):
    e, p, _ = m(strokes, tx, sg, sv)
    e = e.cpu()
    if ref is None: ref = e
    elif not torch.equal(ref, e):
        d = (ref - e).abs(); bad += 1
        print("forward run", i, "differs: max", d.max().item(), "samples", sorted(set(torch.nonzero(d)[:, 0].tolist())), "rows", torch.nonzero(d)[:, 1].min().item(), torch.nonzero(d)[:, 1].max().item(), flush=True)
print("forward nondeterministic runs:", bad)
ref = None; bad = 0
for i in range(10):
    s = dhg_amd.sample(m, tx, sv, L=L, T=4, seed=7).cpu()
    if ref is None: ref = s
    elif not torch.equal(ref, s):
        d = (ref - s).abs(); bad += 1
        print("sample run", i, "differs: max", d.max().item(), "samples", sorted(set(torch.nonzero(d)[:, 0].tolist())), flush=True)
print("sample nondeterministic runs:", bad)
