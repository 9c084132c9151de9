#!/usr/bin/env python3
"""Tape.conv3 / linear / attention at the real shapes against torch autograd (diagnostic, GPU only)."""
import os
import sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_train_model as T  # noqa: E402

g = torch.Generator().manual_seed(1)


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-6)


for B, L, Cin, Cout in ((4, 480, 128, 128), (4, 240, 128, 192), (4, 120, 192, 256), (4, 480, 192, 128), (4, 120, 384, 256), (4, 480, 64, 64), (4, 480, 2, 64)):
    if Cin % 32:
        continue
    x = torch.randn(B * L, Cin, generator=g, requires_grad=True)
    W = torch.randn(Cout, Cin, 3, generator=g, requires_grad=True)
    b = torch.randn(Cout, generator=g, requires_grad=True)
    y, (vx, vW, vb), dy = T._run(lambda t, x, W, b: t.conv3(x, W, b, L), (x, W, b))
    ref = F.conv1d(x.view(B, L, Cin).transpose(1, 2), W, b, padding="same").transpose(1, 2).reshape(B * L, Cout)
    ref.backward(dy)
    print("conv3", B, L, Cin, Cout, "y", rel(y.d, ref), "dx", rel(vx.g, x.grad), "dW", rel(vW.g, W.grad), "db", rel(vb.g, b.grad))
for R, Cin, Cout in ((1920, 128, 128), (1920, 384, 768), (200, 256, 384), (1920, 2, 128), (56, 1280, 256)):
    x = torch.randn(R, Cin, generator=g, requires_grad=True)
    W = torch.randn(Cout, Cin, generator=g, requires_grad=True)
    b = torch.randn(Cout, generator=g, requires_grad=True)
    y, (vx, vW, vb), dy = T._run(lambda t, x, W, b: t.linear(x, W, b), (x, W, b))
    ref = F.linear(x, W, b)
    ref.backward(dy)
    print("linear", R, Cin, Cout, "y", rel(y.d, ref), "dx", rel(vx.g, x.grad), "dW", rel(vW.g, W.grad), "db", rel(vb.g, b.grad))
for B, H, Lq, Lk in ((4, 3, 240, 240), (4, 6, 60, 60), (4, 4, 120, 50), (4, 6, 60, 50)):
    D = 64
    q = torch.randn(B * Lq, H * D, generator=g, requires_grad=True)
    k = torch.randn(B * Lk, H * D, generator=g, requires_grad=True)
    v = torch.randn(B * Lk, H * D, generator=g, requires_grad=True)
    mask = torch.zeros(B, Lk)
    mask[0, Lk - 3:] = 1
    y, (vq, vk, vv), dy = T._run(lambda t, q, k, v: t.attention(q, k, v, B, H, mask.to(T.DEV)), (q, k, v))
    split = lambda a, Ln: a.view(B, Ln, H, D).transpose(1, 2)   # noqa: E731
    ref = F.scaled_dot_product_attention(split(q, Lq), split(k, Lk), split(v, Lk), attn_mask=mask[:, None, None, :] * -1e9)
    ref = ref.transpose(1, 2).reshape(B * Lq, H * D)
    ref.backward(dy)
    print("attn", B, H, Lq, Lk, "y", rel(y.d, ref), "dq", rel(vq.g, q.grad), "dk", rel(vk.g, k.grad), "dv", rel(vv.g, v.grad))
