import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import dhg_amd
from dhg_amd import spec
B, L, Lt = 64, 488, 30
sd = {k: torch.from_numpy(v) for k, v in spec.synthetic_state_dict(2).items()}
m = dhg_amd.DiffusionModel(2, precision="bf16", max_B=B, max_L=L, max_Lt=Lt).eval(); m.load_state_dict(sd)
names = ["enc1", "enc2", "enc3", "enc4", "enc5", "att_layers.0", "att_layers.1", "dec3", "dec2", "dec1"]
tot = 0
for seed in range(12):
    inp = spec.synthetic_inputs(B, L, Lt, seed=100 + seed)
    tx, sv = torch.from_numpy(inp["text"]).cuda(), torch.from_numpy(inp["style"]).cuda()
    strokes = torch.from_numpy(inp["strokes"]).cuda() * (1 + seed)
    sg = torch.full((B, 1), 0.05 + 0.08 * seed).cuda()
    e_full, p_full, _ = m(strokes, tx, sg, sv)
    taps_full = {n: m.debug_read(n).clone() for n in names}
    e_sub, p_sub, _ = m(strokes[40:48].contiguous(), tx[40:48].contiguous(), sg[40:48].contiguous(), sv[40:48].contiguous())
    d = (e_sub - e_full[40:48]).abs().max().item()
    if d > 0:
        tot += 1
        msg = []
        for n in names:
            t = m.debug_read(n)
            tf = taps_full[n].reshape(B, -1)[40:48].reshape(t.shape) if taps_full[n].numel() == t.numel() * 8 else None
            if tf is None: msg.append(f"{n}:shape?{tuple(taps_full[n].shape)}/{tuple(t.shape)}"); continue
            dd = (t - tf).abs()
            msg.append(f"{n}:{dd.max().item():.2e}({int((dd>0).sum())})")
        print("seed", seed, "eps diff", d, " ".join(msg), flush=True)
print("forward shard mismatches:", tot, "of 12")
