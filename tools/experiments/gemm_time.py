#!/usr/bin/env python3
"""dhw_op_gemm timing for the training step's main shapes (batch 32, L=480): 200 back-to-back launches per shape, event-timed.
Diagnostic, GPU only.  Columns: shape, orientation, us per launch, TFLOP/s, fraction of the fp32 MFMA peak (157.3)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dhg_amd  # noqa: E402,F401
from dhg_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.lib()
g = torch.Generator().manual_seed(0)
bf16 = int(os.environ.get("BF16", "0"))
SHAPES = [(1920, 384, 384), (7680, 192, 192), (3840, 256, 256), (15360, 128, 128), (15360, 128, 384), (1920, 768, 384), (1920, 384, 768),
          (384, 384, 1920), (192, 192, 7680), (256, 256, 3840), (128, 128, 15360), (768, 384, 1920)]
for M, N, K in SHAPES:
    forms = ("ATB",) if K > 1000 else ("AB", "ABT")
    for form in forms:
        A = torch.randn(M, K, generator=g).to(dev)
        Bm = torch.randn(K, N, generator=g).to(dev)
        Cm = torch.zeros(M, N, device=dev)
        if form == "ATB":
            As = A.t().contiguous(); sam, sak = 1, M
        else:
            As = A; sam, sak = K, 1
        if form == "ABT":
            Bs = Bm.t().contiguous(); sbk, sbn = 1, K
        else:
            Bs = Bm; sbk, sbn = N, 1
        acc = 1 if form == "ATB" else 0
        d = _lib.GemmDesc(As.data_ptr(), sam, sak, 0, 0, 0, 0, Bs.data_ptr(), sbk, sbn, 0, 0, 0, 0, 0, Cm.data_ptr(), N, 1, 0, 0,
                          M, N, K, 1, 1, 0, 1, None, 1.0, acc, bf16, None)
        for _ in range(5):
            lib.dhw_op_gemm(C.byref(d), None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 200
        e0.record()
        for _ in range(n):
            lib.dhw_op_gemm(C.byref(d), None)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        tf = 2.0 * M * N * K / us / 1e6
        print(f"{M:6d} {N:5d} {K:6d} {form:4s} {us:8.1f} us {tf:7.1f} TFLOP/s  {tf / 157.3:5.3f}")
