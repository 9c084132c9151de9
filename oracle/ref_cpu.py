"""CPU oracle for the reverse-sampling hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch *functional* PyTorch-fp32 CPU restatement of the
reference's denoiser and sampling loop.  It is the checker the HIP path is
compared against; it is NOT part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The product package (``diffusion-handwriting-generation.pytorch_amd``)
never imports anything from ``oracle/`` and fails loudly when the HIP
extension is missing.

Parity pin: every function below is checked against outputs of the *real*
reference (imported in the build container by ``oracle/make_golden.py``) via
the committed fixtures in ``tests/golden/`` (see tests/test_oracle_golden.py).

Reference citations are relative to /root/reference/diffusion_handwriting_generation/.
All functions take a plain ``state_dict`` (name -> fp32 tensor, torch-native
layouts) so that no module classes of the reference are restated.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

LN_EPS = 1e-6  # model.py:25, text_style.py:80


# --------------------------------------------------------------------------
# schedule + step functions (utils/nn.py:19-39, 64-112; inference.py:81-94)
# --------------------------------------------------------------------------
def explin(lo: float, hi: float, n: int) -> torch.Tensor:
    """utils/nn.py:24-39: exp(linspace(ln lo, ln hi, n)) in fp32."""
    return torch.exp(torch.linspace(math.log(lo), math.log(hi), n))


def get_beta_set(T: int = 60) -> torch.Tensor:
    """utils/nn.py:19-21 (T generalised; the reference hard-codes 60)."""
    return 0.02 + explin(1e-5, 0.4, T)


def get_alpha_set(beta_set: torch.Tensor) -> torch.Tensor:
    """inference.py:81."""
    return torch.cumprod(1 - beta_set, dim=0)


def new_diffusion_step(xt, eps, beta, alpha, alpha_next, z):
    """utils/nn.py:110-112 with the randn_like draw made an explicit input."""
    x = (xt - torch.sqrt(1 - alpha) * eps) / torch.sqrt(1 - beta)
    x = x + z * torch.sqrt(1 - alpha_next)
    return x


def standard_diffusion_step(xt, eps, beta, alpha, z=None):
    """utils/nn.py:84-87; z is None <=> add_sigma False."""
    x = (1 / torch.sqrt(1 - beta)) * (xt - (beta * eps / torch.sqrt(1 - alpha)))
    if z is not None:
        x = x + torch.sqrt(beta) * z
    return x


# --------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------
def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def _conv(sd, name, x):
    """Conv1d k=3, 'same' zero padding, dilation 1 (cnn.py:33-47; Appendix C.1)."""
    return F.conv1d(x, sd[name + ".weight"], sd[name + ".bias"], padding="same")


def _ln(x):
    return F.layer_norm(x, (x.shape[-1],), eps=LN_EPS)


def film(sd, name, x, sig):
    """conditioning.py:16-19.  sig: [B,1,32] (or [B,32]); x: [B,T,C]."""
    b = sig.shape[0]
    g = _lin(sd, name + ".gamma_emb", sig).view(b, 1, -1)
    bt = _lin(sd, name + ".beta_emb", sig).view(b, 1, -1)
    return x * g + bt


def ffn(sd, name, x):
    """utils/nn.py:145-175: SiLU -> Linear -> SiLU -> Linear."""
    return _lin(sd, name + ".3", F.silu(_lin(sd, name + ".1", F.silu(x))))


def pos_embeddings(n: int, dim: int, pos_factor) -> torch.Tensor:
    """attention.py:15-23 -> [1, n, dim] fp32, all sines then all cosines."""
    half = dim // 2
    c = math.log(10000) / (half - 1)
    f = torch.exp(torch.arange(half) * -c)
    e = torch.arange(n)[:, None] * f[None, :] * pos_factor
    return torch.cat((e.sin(), e.cos()), dim=-1)[None]


def mha(sd, name, q, k, v, heads: int, mask=None):
    """attention.py:63-87 + 26-46.  mask: [B,1,1,Lk] float, 1 = masked."""
    b, d = q.shape[0], q.shape[-1]
    depth = d // heads
    qh = _lin(sd, name + ".wq", q).view(b, -1, heads, depth).transpose(1, 2)
    kh = _lin(sd, name + ".wk", k).view(b, -1, heads, depth).transpose(1, 2)
    vh = _lin(sd, name + ".wv", v).view(b, -1, heads, depth).transpose(1, 2)
    am = mask * -1e9 if mask is not None else None
    o = F.scaled_dot_product_attention(qh, kh, vh, attn_mask=am)
    o = o.transpose(1, 2).reshape(b, -1, d)
    return _lin(sd, name + ".dense", o)


def conv_block(sd, name, x, sig, taps=None):
    """cnn.py:64-87.  x: [B,C,T] (C-first) -> [B,Cout,T]."""
    skip = _conv(sd, name + ".conv_skip", x)
    h = _conv(sd, name + ".conv1", F.silu(x)).transpose(1, 2)
    h = film(sd, name + ".affine1", h, sig).transpose(1, 2)
    h = _conv(sd, name + ".conv2", F.silu(h)).transpose(1, 2)
    h = film(sd, name + ".affine2", h, sig)
    h = _lin(sd, name + ".fc", F.silu(h))
    h = film(sd, name + ".affine3", h, sig).transpose(1, 2)
    out = h + skip
    if taps is not None:
        taps[name] = out.transpose(1, 2)
    return out


def encoder_layer(sd, name, x, text, sig, mask, heads: int, pos_factor, taps=None):
    """model.py:35-58.  x: [B,T,d], text: [B,Lt,384] -> [B,T,d]."""
    d = x.shape[-1]
    t = _lin(sd, name + ".text_dense", F.silu(text))
    t = film(sd, name + ".affine0", _ln(t), sig)
    t_pe = t + pos_embeddings(t.shape[1], d, 1.0)
    pe_x = pos_embeddings(x.shape[1], d, pos_factor)
    x_pe = x + pe_x
    x2 = mha(sd, name + ".mha", x_pe, t_pe, t, heads, mask)
    x2 = film(sd, name + ".affine1", _ln(x2), sig) + x
    x2_pe = x2 + pe_x
    x3 = mha(sd, name + ".mha2", x2_pe, x2_pe, x2, heads)
    x3 = film(sd, name + ".affine2", _ln(x2 + x3), sig)
    x4 = ffn(sd, name + ".ffn", x3) + x3
    out = film(sd, name + ".affine3", _ln(x4), sig)
    if taps is not None:
        taps[name + ".x2"] = x2
        taps[name + ".x3"] = x3
        taps[name] = out
    return out


def text_style_encoder(sd, name, text, style, sig, taps=None):
    """text_style.py:91-104 (eval: Dropout(0.3) is identity)."""
    b, s, c = style.shape
    st = style.reshape(b, s * 5, c // 5)  # reshape_up(.,5), utils/nn.py:115-127
    st = film(sd, name + ".affine1", _ln(ffn(sd, name + ".style_ffn", st)), sig)
    t = F.embedding(text, sd[name + ".emb.weight"])
    t = film(sd, name + ".affine2", _ln(t), sig)
    m = mha(sd, name + ".mha", t, st, st, 8)
    t = film(sd, name + ".affine3", _ln(t + m), sig)
    out = film(sd, name + ".affine4", _ln(ffn(sd, name + ".text_ffn", t)), sig)
    if taps is not None:
        taps[name + ".style"] = st
        taps[name + ".t2"] = t
        taps[name] = out
    return out


def num_att_layers(sd) -> int:
    n = 0
    while f"att_layers.{n}.text_dense.weight" in sd:
        n += 1
    return n


def forward(sd, strokes, text, sigma, style_vector, taps=None):
    """model.py:121-182.  Returns (eps [B,T,2], pen [B,T]).

    strokes f32 [B,T,2]; text int [B,Lt] (0 = pad); sigma f32 [B,1] or [B,1,1];
    style_vector f32 [B,S,1280].  ``taps`` (optional dict) receives C-last
    copies of block outputs for kernel-level tests.
    """
    text = text.long()
    sig = ffn(sd, "sigma_ffn", sigma)  # model.py:134 (keeps sigma's rank)
    sig = sig.reshape(sig.shape[0], 1, -1)
    mask = torch.eq(text, 0).float()[:, None, None, :]  # utils/nn.py:189-191
    txt = text_style_encoder(sd, "text_style_model", text, style_vector, sig, taps)

    x = _lin(sd, "input_dense", strokes)
    if taps is not None:
        taps["sigma_ffn"] = sig
        taps["input_dense"] = x
    x = x.transpose(1, 2)
    h1 = conv_block(sd, "enc1", x, sig, taps)
    h2 = conv_block(sd, "enc2", F.avg_pool1d(h1, 2), sig, taps)
    h2 = encoder_layer(sd, "enc3", h2.transpose(1, 2), txt, sig, mask, 3, 4, taps).transpose(1, 2)
    h3 = conv_block(sd, "enc4", F.avg_pool1d(h2, 2), sig, taps)
    h3 = encoder_layer(sd, "enc5", h3.transpose(1, 2), txt, sig, mask, 4, 2, taps).transpose(1, 2)
    x = _lin(sd, "att_dense", F.avg_pool1d(h3, 2).transpose(1, 2))
    if taps is not None:
        taps["att_dense"] = x
    for i in range(num_att_layers(sd)):
        x = encoder_layer(sd, f"att_layers.{i}", x, txt, sig, mask, 6, 1.0, taps)
    x = x.transpose(1, 2)

    def up(v):
        return F.interpolate(v, scale_factor=2, mode="nearest")

    def skip(name, h):
        s = _conv(sd, name, h)
        if taps is not None:
            taps[name] = s.transpose(1, 2)
        return s

    x = up(x) + skip("skip_conv3", h3)
    x = conv_block(sd, "dec3", x, sig, taps)
    x = up(x) + skip("skip_conv2", h2)
    x = conv_block(sd, "dec2", x, sig, taps)
    x = up(x) + skip("skip_conv1", h1)
    x = conv_block(sd, "dec1", x, sig, taps)
    x = x.transpose(1, 2)
    eps = _lin(sd, "output_dense", x)
    pen = torch.sigmoid(_lin(sd, "pen_lifts_dense.0", x)).squeeze(-1)
    return eps, pen


def sample(sd, text, style_vector, L: int, noise, T: int = 60, mode: str = "new",
           snapshots=(), grad: bool = False, teacher=None):
    """inference.py:80-96 with the RNG draws replaced by ``noise``.

    noise: f32 [T+1, B, L, 2]; noise[0] = x_T, noise[1 + (T-1-i)] = z drawn at
    loop index i (i = T-1 ... 0), i.e. in consumption order.  In 'standard'
    mode the i == 0 draw is unused (add_sigma = bool(i), inference.py:92).
    Returns (out [B,L,3], {k: x after k steps}).

    teacher = (every, reset [K,B,L,2]): teacher forcing for long schedules — in front of step k*every (k >= 1) the
    state reached so far is recorded (snaps[k*every]) and replaced by reset[k-1]; the reverse process of a random-init
    model gains 1/sqrt(1-beta) per step, so a free-running T=1000 trajectory leaves every meaningful range.
    """
    beta_set = get_beta_set(T)
    alpha_set = get_alpha_set(beta_set)
    bs = text.shape[0]
    x = noise[0].clone()
    snaps = {}
    pen = None
    ctx = torch.enable_grad() if grad else torch.no_grad()
    with ctx:
        for step, i in enumerate(range(T - 1, -1, -1)):
            if teacher is not None and step > 0 and step % teacher[0] == 0:
                snaps[step] = x.detach().clone()
                x = teacher[1][step // teacher[0] - 1].clone()
            alpha = alpha_set[i] * torch.ones((bs, 1, 1))
            beta = beta_set[i] * torch.ones((bs, 1, 1))
            a_next = alpha_set[i - 1] if i > 1 else torch.tensor(1.0)  # inference.py:87
            eps, pen = forward(sd, x, text, torch.sqrt(alpha), style_vector)
            z = noise[1 + step]
            if mode == "standard":
                x = standard_diffusion_step(x, eps, beta, alpha, z if i else None)
            else:
                x = new_diffusion_step(x, eps, beta, alpha, a_next, z)
            if (step + 1) in snapshots:
                snaps[step + 1] = x.detach().clone()
    out = torch.cat((x, pen.unsqueeze(2)), dim=2)  # inference.py:96
    return out.detach(), snaps


# --------------------------------------------------------------------------
# training loss (loss.py:29-37) and the perturbation of train.py:41-44
# --------------------------------------------------------------------------
def loss_fn(eps, score_pred, pen_lifts, pen_lifts_pred, alphas):
    """loss.py:29-37: score_loss = mean(sum((eps - score)^2, -1)); pen loss = mean(mean_L(BCE(pred, clamp(pen))) * abar).
    eps, score_pred [B,L,2]; pen_lifts, pen_lifts_pred [B,L]; alphas [B,1].  Returns (total, score_loss, pen_loss)."""
    score_loss = ((eps - score_pred) ** 2).sum(dim=-1).mean()
    pen = torch.clamp(pen_lifts, min=1e-7, max=1 - 1e-7)
    pen_loss = (F.binary_cross_entropy(pen_lifts_pred, pen, reduction="none").mean(dim=1) * alphas.squeeze(-1)).mean()
    return score_loss + pen_loss, score_loss, pen_loss


def perturb(strokes, eps, alphas):
    """train.py:41-44: x_t = sqrt(abar) * x_0 + sqrt(1 - abar) * eps with abar [B,1] broadcast over [B,L,2]."""
    a = alphas.reshape(-1, 1, 1)
    return torch.sqrt(a) * strokes + torch.sqrt(1 - a) * eps


# --------------------------------------------------------------------------
# tokenizer (tokenizer.py:7-34) and the L heuristic (inference.py:77-78)
# --------------------------------------------------------------------------
_ALPHABET = "_" + "abcdefghijklmnopqrstuvwxyz" + "ABCDEFGHIJKLMNOPQRSTUVWXYZ" + "0123456789" + ".?!,'\"- "


def encode(prompt: str) -> list[int]:
    ids = [(_ALPHABET.index(ch) + 2) if ch in _ALPHABET else 2 for ch in prompt]
    return ids + [1]


def stroke_len(n_tokens: int) -> int:
    t = n_tokens * 16
    return t - (t % 8) + 8
