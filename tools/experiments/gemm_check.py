#!/usr/bin/env python3
"""dhw_op_gemm against torch.matmul over a sweep of shapes / operand orientations (diagnostic, GPU only)."""
import ctypes as C
import itertools
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dhg_amd  # noqa: E402,F401
from dhg_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.lib()
F = 4
bad = 0
g = torch.Generator().manual_seed(0)
for M, N, K in itertools.product((64, 200, 1920, 15360), (64, 128, 384), (2, 50, 60, 96, 128, 192, 384, 480, 1920)):
    for form in ("AB", "ATB", "ABT"):
        for acc in (0, 1):
            A = torch.randn(M, K, generator=g).to(dev)
            Bm = torch.randn(K, N, generator=g).to(dev)
            Cm = torch.randn(M, N, generator=g).to(dev)
            ref = (A.double() @ Bm.double()) + (Cm.double() if acc else 0)
            if form == "ATB":
                As = A.t().contiguous(); sam, sak = 1, M
            else:
                As = A; sam, sak = K, 1
            if form == "ABT":
                Bs = Bm.t().contiguous(); sbk, sbn = 1, K
            else:
                Bs = Bm; sbk, sbn = N, 1
            d = _lib.GemmDesc(As.data_ptr(), sam, sak, 0, 0, 0, 0, Bs.data_ptr(), sbk, sbn, 0, 0, 0, 0, 0, Cm.data_ptr(), N, 1, 0, 0,
                              M, N, K, 1, 1, 0, 1, None, 1.0, acc, 0)
            rc = lib.dhw_op_gemm(C.byref(d), None)
            torch.cuda.synchronize()
            err = float((Cm.double() - ref).abs().max()) / max(float(ref.abs().max()), 1e-6)
            if rc != 0 or err > 1e-5:
                bad += 1
                print("MISMATCH", M, N, K, form, "acc", acc, "rc", rc, "err", err)
print("bad", bad)
