#!/usr/bin/env python3
"""Isolated timings of the training step's strided fp32 GEMM (dhw_op_gemm) on the shapes the model issues most:
python tools/bench_sgemm.py  -> one line per shape: us per call, TFLOP/s (of the 157 TFLOP/s fp32-MFMA peak)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dhg_amd import train_model as tm  # noqa: E402


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    dev = torch.device("cuda", 0)
    t = tm.Tape(dev)
    for name, R, K, N, L in (("linear 3840x256x256", 3840, 256, 256, 0), ("linear 1920x384x768", 1920, 384, 768, 0), ("linear 1920x768x384", 1920, 768, 384, 0),
                             ("linear 7680x192x192", 7680, 192, 192, 0), ("conv 15360x(3x128)x128", 15360, 128, 128, 480), ("conv 7680x(3x256)x192", 7680, 256, 192, 240),
                             ("conv 3840x(3x384)x256", 3840, 384, 256, 120)):
        x = tm.Var(torch.randn(R, K, device=dev))
        W = tm.Var(torch.randn((N, K, 3) if L else (N, K), device=dev))
        b = tm.Var(torch.randn(N, device=dev))
        fwd = (lambda: t.conv3(x, W, b, L)) if L else (lambda: t.linear(x, W, b))
        flops = 2 * R * K * N * (3 if L else 1)
        y = fwd()
        y.g = torch.randn_like(y.d)
        bwd = t.steps[-1]
        us_f = timed(fwd)
        us_b = timed(bwd)
        t.join()
        t.steps.clear()
        print(f"{name:26s} fwd {us_f:7.1f} us {flops / us_f / 1e6:6.1f} TF | dgrad + wgrad + bias {us_b:7.1f} us {2 * flops / us_b / 1e6:6.1f} TF", flush=True)


if __name__ == "__main__":
    main()
