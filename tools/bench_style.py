#!/usr/bin/env python3
"""Time the StyleExtractor front end (dhg_amd.StyleExtractor): B grey line images [B,1,96,W] -> [B,14,1280],
random-init MobileNetV2 weights.  python tools/bench_style.py [--batch 8] [--width 1400] [--precision fp32]"""
import argparse
import json
import os
import sys
import time
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import dhg_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--width", type=int, default=1400)
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ex = dhg_amd.StyleExtractor(None, precision=a.precision)
    img = np.random.default_rng(0).integers(0, 256, size=(a.batch, 1, 96, a.width)).astype(np.float32)
    out = ex.forward(img)
    torch.cuda.synchronize()
    x = torch.from_numpy(img).cuda()
    ex.forward(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = ex.forward(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(json.dumps({"metric": "StyleExtractor images/s (96 x %d grey lines, device-resident input)" % a.width, "value": round(a.batch / dt, 1),
                      "ms_per_batch": round(dt * 1e3, 3), "batch": a.batch, "precision": a.precision, "finite": bool(torch.isfinite(out).all())}))


if __name__ == "__main__":
    main()
