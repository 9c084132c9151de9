#!/usr/bin/env python3
"""Bring-up check on the GPU box: per-block error of the HIP path against the golden taps /
forwards (tests/golden), for both precisions.  Prints a table; exit code 0 always (diagnostic)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dhg_amd  # noqa: E402
from dhg_amd import spec  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def model(nl, prec):
    m = dhg_amd.DiffusionModel(nl, precision=prec, max_B=8, max_L=488, max_Lt=40)
    sd = {k: torch.from_numpy(v) for k, v in spec.synthetic_state_dict(nl).items()}
    m.load_state_dict(sd, strict=True)
    return m.eval()


def main():
    alpha = dhg_amd.get_alpha_set()
    for prec in ("fp32", "bf16"):
        print("=" * 20, prec)
        m = model(2, prec)
        g = np.load(os.path.join(G, "taps.npz"))
        B, L, Lt = int(g["B"]), int(g["L"]), int(g["Lt"])
        inp = spec.synthetic_inputs(B, L, Lt, seed=int(g["seed"]), pad=int(g["pad"]))
        sg = torch.sqrt(alpha[int(g["sigma_index"])]) * torch.ones((B, 1, 1))
        t0 = time.time()
        eps, pen, _ = m(torch.from_numpy(inp["strokes"]).cuda(), torch.from_numpy(inp["text"]).cuda(), sg.cuda(),
                        torch.from_numpy(inp["style"]).cuda())
        torch.cuda.synchronize()
        print("first forward %.2fs" % (time.time() - t0))
        order = ["sigma_ffn", "text_style_model", "input_dense", "enc1", "enc2", "enc3", "enc4", "enc5", "att_dense",
                 "att_layers.0", "att_layers.1", "dec3", "dec2", "dec1"]
        for name in order:
            ref = g["tap_" + name]
            got = m.debug_read(name).numpy()
            if name == "sigma_ffn":
                got = got.reshape(ref.shape)
            err = np.abs(got - ref)
            print(f"{name:20s} shape {str(ref.shape):16s} max|ref| {np.abs(ref).max():9.4f} maxerr {err.max():.3e} "
                  f"meanerr {err.mean():.3e} nan {int(np.isnan(got).sum())}")
        print(f"{'eps':20s} maxerr {np.abs(eps.cpu().numpy() - g['eps']).max():.3e}")
        print(f"{'pen':20s} maxerr {np.abs(pen.cpu().numpy() - g['pen']).max():.3e}")
        for fname in ("fwd_main.npz", "fwd_pad.npz", "fwd_s1.npz"):
            g = np.load(os.path.join(G, fname))
            B, L, Lt, S = int(g["B"]), int(g["L"]), int(g["Lt"]), int(g["S"])
            inp = spec.synthetic_inputs(B, L, Lt, S=S, seed=int(g["seed"]), pad=int(g["pad"]))
            if "text" in g.files:
                inp["text"] = g["text"]
            for i in (59, 30, 0):
                sg = torch.sqrt(alpha[i]) * torch.ones((B, 1, 1))
                eps, pen, _ = m(torch.from_numpy(inp["strokes"]).cuda(), torch.from_numpy(inp["text"]).cuda(), sg.cuda(),
                                torch.from_numpy(inp["style"]).cuda())
                print(f"{fname} i={i:2d} eps maxerr {np.abs(eps.cpu().numpy() - g[f'eps_i{i}']).max():.3e} "
                      f"pen maxerr {np.abs(pen.cpu().numpy() - g[f'pen_i{i}']).max():.3e}")
        for fname, mode in (("loop_new.npz", "new"), ("loop_std.npz", "standard")):
            g = np.load(os.path.join(G, fname))
            B, L, Lt = int(g["B"]), int(g["L"]), int(g["Lt"])
            inp = spec.synthetic_inputs(B, L, Lt, seed=int(g["seed"]))
            t0 = time.time()
            out = dhg_amd.sample(m, torch.from_numpy(inp["text"]).cuda(), torch.from_numpy(inp["style"]).cuda(), L=L,
                                 diffusion_mode=mode, noise=torch.from_numpy(inp["noise"]).cuda())
            torch.cuda.synchronize()
            dt = time.time() - t0
            o = out.cpu().numpy()
            bits = np.round(o[..., 2]).astype(np.uint8)
            print(f"{fname} {dt:.2f}s x maxerr {np.abs(o[..., :2] - g['out'][..., :2]).max():.3e} (max|x| {np.abs(g['out'][..., :2]).max():.1f}) "
                  f"pen maxerr {np.abs(o[..., 2] - g['out'][..., 2]).max():.3e} pen flips {int((bits != g['pen_bits']).sum())}/{bits.size}")
        # throughput probe
        B, L, Lt = 8, 488, 30
        inp = spec.synthetic_inputs(B, L, Lt, seed=5)
        tx, sv = torch.from_numpy(inp["text"]).cuda(), torch.from_numpy(inp["style"]).cuda()
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.time()
            dhg_amd.sample(m, tx, sv, L=L, seed=rep)
            torch.cuda.synchronize()
            print(f"sample B={B} L={L} T=60: {time.time() - t0:.3f}s")


if __name__ == "__main__":
    main()
