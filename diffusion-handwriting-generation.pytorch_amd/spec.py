"""Parameter inventory of the denoiser and a portable synthetic-weight generator.

The weight interchange format of the hot path is the reference's
``state_dict`` (323 tensors for num_layers=2, torch-native layouts; SURVEY
§8(b); reference model.py:64-119, cnn.py:24-50, text_style.py:65-89,
conditioning.py:9-14).  ``param_spec`` lists (name, shape, kind) in the
reference's registration order so that a checkpoint of the reference loads by
key name.  ``synthetic_state_dict`` fills it from numpy PCG64 streams keyed by
the *tensor name*, so every machine regenerates bit-identical weights without
the reference and without torch's RNG.
"""
from __future__ import annotations

import zlib

import numpy as np

SIGMA_DIM = 32       # conditioning.py:9-10 hard-codes 32 (= c1 // 4)
SIGMA_HIDDEN = 2048  # model.py:83
VOCAB = 73           # text_style.py:71
STYLE_CH = 256       # 1280 / 5, text_style.py:74


def _affine(name, c):
    return [
        (f"{name}.gamma_emb.weight", (c, SIGMA_DIM), "linear_w"),
        (f"{name}.gamma_emb.bias", (c,), "ones"),
        (f"{name}.beta_emb.weight", (c, SIGMA_DIM), "linear_w"),
        (f"{name}.beta_emb.bias", (c,), "linear_b:%d" % SIGMA_DIM),
    ]


def _linear(name, cin, cout):
    return [(f"{name}.weight", (cout, cin), "linear_w"), (f"{name}.bias", (cout,), "linear_b:%d" % cin)]


def _conv(name, cin, cout):
    return [(f"{name}.weight", (cout, cin, 3), "conv_w"), (f"{name}.bias", (cout,), "linear_b:%d" % (cin * 3))]


def _conv_block(name, cin, cout):
    out = _affine(f"{name}.affine1", cout // 2) + _affine(f"{name}.affine2", cout) + _affine(f"{name}.affine3", cout)
    out += _conv(f"{name}.conv_skip", cin, cout) + _conv(f"{name}.conv1", cin, cout // 2)
    out += _conv(f"{name}.conv2", cout // 2, cout) + _linear(f"{name}.fc", cout, cout)
    return out


def _mha(name, d):
    return _linear(f"{name}.wq", d, d) + _linear(f"{name}.wk", d, d) + _linear(f"{name}.wv", d, d) + _linear(f"{name}.dense", d, d)


def _encoder_layer(name, d_inp, d):
    out = _linear(f"{name}.text_dense", d_inp, d)
    out += _linear(f"{name}.ffn.1", d, 2 * d) + _linear(f"{name}.ffn.3", 2 * d, d)
    out += _mha(f"{name}.mha", d) + _mha(f"{name}.mha2", d)
    for k in range(4):
        out += _affine(f"{name}.affine{k}", d)
    return out


def param_spec(num_layers: int = 4, c1: int = 128, c2: int = 192, c3: int = 256):
    """[(state_dict key, shape, init kind)] in the reference's key order."""
    dt = 2 * c2  # text / bottleneck width
    s = _linear("input_dense", 2, c1)
    s += _linear("sigma_ffn.1", 1, SIGMA_HIDDEN) + _linear("sigma_ffn.3", SIGMA_HIDDEN, c1 // 4)
    s += _conv_block("enc1", c1, c1) + _conv_block("enc2", c1, c2)
    s += _encoder_layer("enc3", dt, c2)
    s += _conv_block("enc4", c2, c3)
    s += _encoder_layer("enc5", dt, c3)
    s += _conv("skip_conv1", c1, c2) + _conv("skip_conv2", c2, c3) + _conv("skip_conv3", c3, dt)
    t = "text_style_model"
    s += [(f"{t}.emb.weight", (VOCAB, dt), "normal")]
    s += _linear(f"{t}.style_ffn.1", STYLE_CH, 4 * c2) + _linear(f"{t}.style_ffn.3", 4 * c2, dt)
    s += _linear(f"{t}.text_ffn.1", dt, 2 * dt) + _linear(f"{t}.text_ffn.3", 2 * dt, dt)
    s += _mha(f"{t}.mha", dt)
    for k in range(1, 5):
        s += _affine(f"{t}.affine{k}", dt)
    s += _linear("att_dense", 2 * c1, dt)
    for i in range(num_layers):
        s += _encoder_layer(f"att_layers.{i}", dt, dt)
    s += _conv_block("dec3", dt, c3) + _conv_block("dec2", c3, c2) + _conv_block("dec1", c2, c1)
    s += _linear("output_dense", c1, 2) + _linear("pen_lifts_dense.0", c1, 1)
    return s


def _fill(name: str, shape, kind: str, seed: int) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))
    if kind == "ones":
        return np.ones(shape, np.float32)
    if kind == "normal":
        return rng.standard_normal(shape, dtype=np.float32)
    if kind in ("linear_w", "conv_w"):
        fan_in = int(np.prod(shape[1:]))
    else:  # "linear_b:<fan_in>"
        fan_in = int(kind.split(":")[1])
    bound = 1.0 / np.sqrt(fan_in)
    return rng.uniform(-bound, bound, size=shape).astype(np.float32)


def synthetic_state_dict(num_layers=2, c1=128, c2=192, c3=256, seed=0):
    """name -> fp32 ndarray; torch-default-like init scale."""
    return {n: _fill(n, shp, kind, seed) for n, shp, kind in param_spec(num_layers, c1, c2, c3)}


def synthetic_inputs(B: int, L: int, Lt: int, S: int = 14, seed: int = 1, pad: int = 0, T: int = 60):
    """Deterministic synthetic sampler inputs.

    Returns dict(text int64 [B,Lt] in [1,72] with ``pad`` trailing zeros,
    style f32 [B,S,1280], strokes f32 [B,L,2], noise f32 [T+1,B,L,2]).
    Each sample's streams are keyed by its *global* index so a sharded run
    (rank r gets samples r*B ... r*B+B-1 via ``first``) sees identical data.
    """
    return synthetic_inputs_range(0, B, L, Lt, S, seed, pad, T)


def synthetic_inputs_range(first: int, B: int, L: int, Lt: int, S: int = 14, seed: int = 1, pad: int = 0, T: int = 60):
    text = np.zeros((B, Lt), np.int64)
    style = np.zeros((B, S, 1280), np.float32)
    strokes = np.zeros((B, L, 2), np.float32)
    noise = np.zeros((T + 1, B, L, 2), np.float32)
    for b in range(B):
        rng = np.random.Generator(np.random.PCG64([seed, first + b]))
        text[b] = rng.integers(1, 73, size=Lt)
        if pad:
            text[b, Lt - pad:] = 0
        style[b] = rng.standard_normal((S, 1280), dtype=np.float32)
        strokes[b] = rng.standard_normal((L, 2), dtype=np.float32)
        noise[:, b] = rng.standard_normal((T + 1, L, 2), dtype=np.float32)
    return {"text": text, "style": style, "strokes": strokes, "noise": noise}
