#!/usr/bin/env python3
"""Generate tests/golden/fwd_c2_*.npz: the REAL reference built with c2 != 192 (model.py:64-71 accepts any c2 divisible by 12).

TEST INFRASTRUCTURE, same rules as make_golden.py: runs only where /root/reference exists, imports the reference unmodified,
stores data only (shapes, seeds, the reference's outputs); weights and inputs are regenerated from seeds by spec.py.

  fwd_c2_48.npz   c = (128, 48, 256), num_layers 2, B=2 L=136 Lt=12 (3 pad tokens): eps / pen at sigma = sqrt(abar_i), i in {59, 30, 0},
                  and at per-row sigma; plus x after a 6-step reverse loop (new_diffusion_step) from the shared noise stream
  fwd_c2_96.npz   c = (128, 96, 256), same
  fwd_c2_24.npz   c = (128, 24, 256), B=1 L=64 Lt=8
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    spec = mg._load_spec()
    DiffusionModel, _, refnn = mg._import_reference()
    beta = refnn.get_beta_set()
    alpha = torch.cumprod(1 - beta, dim=0)

    def build(c2, nl=2, seed=0):
        m = DiffusionModel(nl, 128, c2, 256)
        sd = spec.synthetic_state_dict(nl, 128, c2, 256, seed=seed)
        ref_sd = m.state_dict()
        assert list(ref_sd.keys()) == [n for n, _, _ in spec.param_spec(nl, 128, c2, 256)]
        for k, v in ref_sd.items():
            assert tuple(v.shape) == sd[k].shape, k
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        return m.eval()

    def fwd(m, x, text, sigma, style):
        with torch.no_grad():
            eps, pen, _ = m(x, text, sigma, style)
        return eps, pen

    for c2, B, L, Lt, pad, seed in ((48, 2, 136, 12, 3, 41), (96, 2, 136, 12, 3, 42), (24, 1, 64, 8, 0, 43)):
        m = build(c2)
        T = 6
        inp = spec.synthetic_inputs(B, L, Lt, seed=seed, pad=pad, T=T)
        x0, text, style = (torch.from_numpy(inp[k]) for k in ("strokes", "text", "style"))
        out = {"c2": c2, "B": B, "L": L, "Lt": Lt, "pad": pad, "seed": seed, "T": T}
        for i in (59, 30, 0):
            e, p = fwd(m, x0, text, torch.sqrt(alpha[i]) * torch.ones((B, 1, 1)), style)
            out[f"eps_i{i}"], out[f"pen_i{i}"] = e.numpy(), p.numpy()
        rng = np.random.Generator(np.random.PCG64([seed, 777]))
        sg = torch.from_numpy(rng.uniform(0.1, 1.0, size=(B, 1)).astype(np.float32))
        e, p = fwd(m, x0, text, sg, style)
        out["sigma_rand"], out["eps_rand"], out["pen_rand"] = sg.numpy(), e.numpy(), p.numpy()
        # a short reverse loop over the LAST T entries' worth of a T-step schedule (inference.py:80-96 with diffusion_mode "new")
        bs = 0.02 + torch.exp(torch.linspace(np.log(1e-5), np.log(0.4), T))   # utils/nn.py:19-39 at T points
        al = torch.cumprod(1 - bs, dim=0)
        noise = torch.from_numpy(inp["noise"])
        x = noise[0].clone()
        for i in range(T - 1, -1, -1):
            a = al[i] * torch.ones((B, 1, 1))
            b = bs[i] * torch.ones((B, 1, 1))
            a_next = al[i - 1] if i > 1 else torch.tensor(1.0)
            e, pen = fwd(m, x, text, torch.sqrt(a), style)
            draws = iter([noise[1 + (T - 1 - i)]])
            orig = torch.randn_like
            torch.randn_like = lambda t: next(draws)
            try:
                x = refnn.new_diffusion_step(x, e, b, a, a_next).detach()
            finally:
                torch.randn_like = orig
        out["loop_out"] = torch.cat((x, pen.unsqueeze(2)), dim=2).numpy()
        np.savez(os.path.join(mg.OUT, f"fwd_c2_{c2}.npz"), **out)
        print(c2, "eps", float(np.abs(out["eps_i30"]).max()), "loop", float(np.abs(out["loop_out"]).max()))


if __name__ == "__main__":
    main()
