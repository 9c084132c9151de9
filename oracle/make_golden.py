#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference in this container.

TEST INFRASTRUCTURE.  Runs only where /root/reference exists (the build
container); the GPU box only ever sees the committed fixtures.  The reference
is imported unmodified; ``torchvision`` (absent here, used only inside
StyleExtractor.__init__, text_style.py:19-22) is satisfied by an empty
placeholder module so that model.py / utils/nn.py / tokenizer.py import.

Fixtures hold *data only*: seeds + small input descriptors + the reference's
outputs.  Weights and inputs are regenerated from seeds by
``<package>/spec.py`` (numpy PCG64 keyed by tensor name), never stored.

  sched.npz      beta_set[60], alpha_set[60]                 (nn.py:19-39, inference.py:81)
  fwd_main.npz   B=2 L=488 Lt=30: eps/pen at sigma=sqrt(abar_i), i in {59,30,0}, and per-row sigma
  fwd_pad.npz    same shape, 5 trailing pad tokens            (mask path nn.py:189, attention.py:44)
  fwd_s1.npz     B=2 L=400 Lt=40 S=1, 0/1 tokens              (tests/test_model.py:14-21 shapes)
  fwd_nl4.npz    num_layers=4 (class default model.py:66), B=1 L=64 Lt=8
  taps.npz       B=1 L=136 Lt=12 (3 pad): every top-level block's output (C-last)
  loop_new.npz / loop_std.npz  B=2 L=488 T=60: x after {1,10,30,60} steps, final [B,L,3]
  keys.json      the reference's state_dict keys + shapes for num_layers 2 and 4
  tokenizer.json known answers of Tokenizer.encode + the L heuristic (inference.py:72-78)
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def _load_spec():
    p = os.path.join(ROOT, "diffusion-handwriting-generation.pytorch_amd", "spec.py")
    s = importlib.util.spec_from_file_location("dhw_spec", p)
    m = importlib.util.module_from_spec(s)
    s.loader.exec_module(m)
    return m


def _import_reference():
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tv.models = tvm
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.models", tvm)
    sys.path.insert(0, REF)
    from diffusion_handwriting_generation.model import DiffusionModel
    from diffusion_handwriting_generation.tokenizer import Tokenizer
    from diffusion_handwriting_generation.utils import nn as refnn
    return DiffusionModel, Tokenizer, refnn


def build_ref_model(DiffusionModel, spec, num_layers, seed=0):
    m = DiffusionModel(num_layers, 128, 192, 256)
    sd = spec.synthetic_state_dict(num_layers, 128, 192, 256, seed=seed)
    ref_sd = m.state_dict()
    assert list(ref_sd.keys()) == [n for n, _, _ in spec.param_spec(num_layers)], "state_dict key order differs"
    for k, v in ref_sd.items():
        assert tuple(v.shape) == sd[k].shape, k
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m.eval()
    return m


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    spec = _load_spec()
    DiffusionModel, Tokenizer, refnn = _import_reference()

    # ---- state_dict inventory of the reference (names, shapes, order)
    keys = {}
    for nl in (2, 4):
        keys[str(nl)] = [[k, list(v.shape)] for k, v in DiffusionModel(nl, 128, 192, 256).state_dict().items()]
    with open(os.path.join(OUT, "keys.json"), "w") as f:
        json.dump(keys, f)
    if "--keys-only" in sys.argv:
        return

    # ---- schedule
    beta = refnn.get_beta_set()
    alpha = torch.cumprod(1 - beta, dim=0)
    np.savez(os.path.join(OUT, "sched.npz"), beta=beta.numpy(), alpha=alpha.numpy())

    model = build_ref_model(DiffusionModel, spec, 2)

    def fwd(m, inp, sigma):
        with torch.no_grad():
            eps, pen, none = m(torch.from_numpy(inp["strokes"]), torch.from_numpy(inp["text"]),
                               sigma, torch.from_numpy(inp["style"]))
        assert none is None
        return eps.numpy(), pen.numpy()

    # ---- single forwards
    def fwd_cases(m, B, L, Lt, S, seed, pad, fname, text_override=None, extra=None):
        inp = spec.synthetic_inputs(B, L, Lt, S=S, seed=seed, pad=pad)
        if text_override is not None:
            inp["text"] = text_override
        out = {"B": B, "L": L, "Lt": Lt, "S": S, "seed": seed, "pad": pad}
        if text_override is not None:
            out["text"] = text_override
        for i in (59, 30, 0):
            sg = torch.sqrt(alpha[i]) * torch.ones((B, 1, 1))
            e, p = fwd(m, inp, sg)
            out[f"eps_i{i}"], out[f"pen_i{i}"] = e, p
        rng = np.random.Generator(np.random.PCG64([seed, 777]))
        sg = torch.from_numpy(rng.uniform(0.1, 1.0, size=(B, 1)).astype(np.float32))  # training-style [B,1]
        e, p = fwd(m, inp, sg)
        out["sigma_rand"], out["eps_rand"], out["pen_rand"] = sg.numpy(), e, p
        if extra:
            out.update(extra)
        np.savez(os.path.join(OUT, fname), **out)

    fwd_cases(model, 2, 488, 30, 14, 11, 0, "fwd_main.npz")
    fwd_cases(model, 2, 488, 30, 14, 12, 5, "fwd_pad.npz")
    rng = np.random.Generator(np.random.PCG64(13))
    t01 = (rng.uniform(size=(2, 40)) < 0.25).astype(np.int64)  # tests/test_model.py:18
    fwd_cases(model, 2, 400, 40, 1, 13, 0, "fwd_s1.npz", text_override=t01)
    model4 = build_ref_model(DiffusionModel, spec, 4)
    fwd_cases(model4, 1, 64, 8, 14, 14, 2, "fwd_nl4.npz")

    # ---- block taps via forward hooks on the reference's own modules
    B, L, Lt, pad, seed = 1, 136, 12, 3, 21
    inp = spec.synthetic_inputs(B, L, Lt, seed=seed, pad=pad)
    taps = {}

    def hook(name, clast):
        def f(mod, args, out):
            o = out[0] if isinstance(out, tuple) else out
            taps[name] = (o if clast else o.transpose(1, 2)).detach().numpy().copy()
        return f

    hs = []
    for n in ("enc1", "enc2", "enc4", "dec3", "dec2", "dec1"):
        hs.append(getattr(model, n).register_forward_hook(hook(n, False)))
    for n in ("enc3", "enc5", "text_style_model", "att_dense", "sigma_ffn", "input_dense"):
        hs.append(getattr(model, n).register_forward_hook(hook(n, True)))
    for i, lyr in enumerate(model.att_layers):
        hs.append(lyr.register_forward_hook(hook(f"att_layers.{i}", True)))
    for n in ("skip_conv1", "skip_conv2", "skip_conv3"):
        hs.append(getattr(model, n).register_forward_hook(hook(n, False)))
    sg = torch.sqrt(alpha[30]) * torch.ones((B, 1, 1))
    e, p = fwd(model, inp, sg)
    for h in hs:
        h.remove()
    np.savez(os.path.join(OUT, "taps.npz"), B=B, L=L, Lt=Lt, pad=pad, seed=seed, sigma_index=30,
             eps=e, pen=p, **{"tap_" + k: v for k, v in taps.items()})

    # ---- full 60-step loops (inference.py:80-96 replayed around the imported model/step functions)
    def loop(mode, fname, B=2, L=488, Lt=30, seed=31):
        inp = spec.synthetic_inputs(B, L, Lt, seed=seed)
        text = torch.from_numpy(inp["text"])
        style = torch.from_numpy(inp["style"])
        noise = torch.from_numpy(inp["noise"])
        beta_set = refnn.get_beta_set()
        alpha_set = torch.cumprod(1 - beta_set, dim=0)
        x = noise[0].clone()
        snaps = {}
        # torch.randn_like is the reference's noise source; feed the pre-generated stream instead
        draws = iter(noise[1:])
        orig = torch.randn_like
        torch.randn_like = lambda t: next(draws)
        try:
            for step, i in enumerate(range(len(beta_set) - 1, -1, -1)):
                a = alpha_set[i] * torch.ones((B, 1, 1))
                b = beta_set[i] * torch.ones((B, 1, 1))
                a_next = alpha_set[i - 1] if i > 1 else torch.tensor(1.0)
                model_out, pen_lifts, _ = model(x, text, torch.sqrt(a), style)
                if mode == "standard":
                    if not i:
                        next(draws)  # keep noise[1+step] <-> loop index alignment
                    x = refnn.standard_diffusion_step(x, model_out, b, a, add_sigma=bool(i))
                else:
                    x = refnn.new_diffusion_step(x, model_out, b, a, a_next)
                x = x.detach()
                if step + 1 in (1, 10, 30, 60):
                    snaps[f"x_after_{step + 1}"] = x.numpy().copy()
        finally:
            torch.randn_like = orig
        out = torch.cat((x, pen_lifts.unsqueeze(2)), dim=2).detach().numpy()
        np.savez(os.path.join(OUT, fname), B=B, L=L, Lt=Lt, seed=seed, mode=mode, out=out,
                 pen_bits=np.round(out[..., 2]).astype(np.uint8), **snaps)

    loop("new", "loop_new.npz")
    loop("standard", "loop_std.npz")

    # ---- tokenizer known answers
    tk = Tokenizer()
    prompts = ["Follow the White Rabbit", "a" * 29, "Hello, World! 123?", "~unknown^chars~", ""]
    ka = []
    for pr in prompts:
        ids = [int(v) for v in tk.encode(pr)]
        ts = len(ids) * 16
        ts = ts - (ts % 8) + 8
        ka.append({"prompt": pr, "ids": ids, "L": ts})
    with open(os.path.join(OUT, "tokenizer.json"), "w") as f:
        json.dump(ka, f, indent=1)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
