#!/usr/bin/env python3
"""Measures (on the GPU box) the quantities the parity tests' tolerances are set from: bf16 tap errors against the
reference taps, switch / tile-variant self-consistency in bf16, pen-bit flips of the bf16 path against the fp32 path at the
bench size (B=64, T=60, L=488, same external noise), and the long-schedule configuration (configs[3]).
Writes gpurun_out/parity_measure.json."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import dhg_amd
from dhg_amd import spec

res = {}
sd = lambda nl=2: {k: torch.from_numpy(v) for k, v in spec.synthetic_state_dict(nl).items()}
GOLD = os.path.join(ROOT, "tests", "golden")

def model(prec, B, L, Lt, sdict=None):
    m = dhg_amd.DiffusionModel(2, precision=prec, max_B=B, max_L=L, max_Lt=Lt).eval()
    m.load_state_dict(sdict or sd())
    return m

# 1. taps
g = np.load(os.path.join(GOLD, "taps.npz"))
B, L, Lt = int(g["B"]), int(g["L"]), int(g["Lt"])
inp = spec.synthetic_inputs(B, L, Lt, seed=int(g["seed"]), pad=int(g["pad"]))
alpha = dhg_amd.get_alpha_set()
taps = {}
for prec in ("fp32", "bf16"):
    m = model(prec, 8, 488, 40)
    m(torch.from_numpy(inp["strokes"]).cuda(), torch.from_numpy(inp["text"]).cuda(),
      (torch.sqrt(alpha[int(g["sigma_index"])]) * torch.ones((B, 1, 1))).cuda(), torch.from_numpy(inp["style"]).cuda())
    for f in g.files:
        if not f.startswith("tap_"):
            continue
        name = f[4:]
        try:
            got = m.debug_read(name).numpy().reshape(g[f].shape)
        except Exception as e:
            taps.setdefault(name, {})[prec] = "n/a: " + str(e)[:60]
            continue
        taps.setdefault(name, {})[prec] = float(np.abs(got - g[f]).max())
        taps[name]["ref_absmax"] = float(np.abs(g[f]).max())
    del m
res["tap_err"] = taps
print(json.dumps(taps, indent=1), flush=True)

# 2. pen flips at the bench size
B, L, Lt, T = 64, 488, 30, 60
inp = spec.synthetic_inputs(B, L, Lt, seed=31, T=T)
tx, sv, nz = (torch.from_numpy(inp[k]).cuda() for k in ("text", "style", "noise"))
outs = {}
for prec in ("fp32", "bf16"):
    m = model(prec, B, L, Lt)
    outs[prec] = dhg_amd.sample(m, tx, sv, L=L, T=T, noise=nz).cpu().numpy()
    del m
ref, got = outs["fp32"], outs["bf16"]
bits_r, bits_g = np.round(ref[..., 2]), np.round(got[..., 2])
fl = bits_r != bits_g
res["pen_flips_bench"] = {"B": B, "T": T, "L": L, "flipped": int(fl.sum()), "total": int(fl.size),
                          "max_abs_p_minus_half_among_flips": float(np.abs(ref[..., 2][fl] - 0.5).max()) if fl.any() else 0.0,
                          "max_abs_dp": float(np.abs(ref[..., 2] - got[..., 2]).max()),
                          "traj_rel_err": float(np.abs(ref[..., :2] - got[..., :2]).max() / np.abs(ref[..., :2]).max()),
                          "p_min": float(ref[..., 2].min()), "p_max": float(ref[..., 2].max()),
                          "frac_p_within_0.005_of_half": float((np.abs(ref[..., 2] - 0.5) < 0.005).mean())}
print(json.dumps(res["pen_flips_bench"]), flush=True)

# 3. configs[3]: long schedule with the output head scaled so the random-init trajectory stays finite
def scaled_sd(scale):
    d = sd()
    d["output_dense.weight"] = d["output_dense.weight"] * scale
    d["output_dense.bias"] = d["output_dense.bias"] * scale
    return d
B, L, Lt, T = 32, 1000, 62, 1000
inp = spec.synthetic_inputs(B, L, Lt, seed=5, T=0)
tx, sv = torch.from_numpy(inp["text"]).cuda(), torch.from_numpy(inp["style"]).cuda()
c3 = {}
for scale in (1.0, 0.05, 0.01):
    m = model("bf16", B, L, Lt, scaled_sd(scale))
    out = dhg_amd.sample(m, tx, sv, L=L, T=T, seed=3)
    torch.cuda.synchronize(); t0 = time.time()
    out = dhg_amd.sample(m, tx, sv, L=L, T=T, seed=3)
    torch.cuda.synchronize(); dt = time.time() - t0
    c3[str(scale)] = {"ms": dt * 1e3, "finite": bool(torch.isfinite(out).all()), "max_abs_x": float(out[..., :2].abs().max())}
    print(scale, c3[str(scale)], flush=True)
    del m
res["config3"] = c3
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "parity_measure.json"), "w"), indent=1)
