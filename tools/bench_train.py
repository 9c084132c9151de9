#!/usr/bin/env python3
"""Time the native training step (dhg_amd.train_model.train_step) at BASELINE configs[4]'s per-GPU shape: batch 32 (global 256
over 8 GPUs), L = 480 strokes, 50 text tokens (configs/best.yml dataset_args), synthetic batch, random-init weights.
  python tools/bench_train.py [--batch 32] [--steps 10] [--warmup 2]
Prints one JSON line: ms per update, samples/s, kernel launches per update."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dhg_amd import spec, train, train_model as tm  # noqa: E402


def main():
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--L", type=int, default=480)
    ap.add_argument("--Lt", type=int, default=50)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--eager", action="store_true", help="launch every kernel from the host instead of replaying the captured hipGraph")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    sd = spec.synthetic_state_dict(2, 128, 192, 256, seed=0)
    model = tm.TrainModel(sd, num_layers=2, device=dev)
    opt = train.Adam(model.parameters())
    inp = spec.synthetic_inputs(a.batch, a.L, a.Lt, S=14, seed=3, pad=5)
    g = torch.Generator().manual_seed(3)
    strokes3 = torch.cat([torch.from_numpy(inp["strokes"]), (torch.rand(a.batch, a.L, 1, generator=g) < 0.1).float()], dim=-1)
    batch = {"strokes": strokes3, "text": torch.from_numpy(inp["text"]), "style": torch.from_numpy(inp["style"])}
    beta = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "sched.npz"))
    alpha_set = torch.from_numpy(beta["alpha"])
    graphed = tm.GraphedTrainStep(model, opt, a.batch, a.L, a.Lt)
    losses = []
    for step in range(1, a.warmup + a.steps + 1):
        if step == a.warmup + 1:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        out = graphed(batch, alpha_set, step, graph=not a.eager)
        losses.append(out.clone())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(json.dumps({"metric": "training updates (forward + loss + backward + clip + Adam)", "ms_per_update": round(dt * 1e3, 2),
                      "samples_per_s": round(a.batch / dt, 1), "batch": a.batch, "L": a.L, "Lt": a.Lt, "dtype": "f32", "launch": "eager" if a.eager else "hipGraph",
                      "loss_first_last": [float(losses[0][0]), float(losses[-1][0])]}))


if __name__ == "__main__":
    main()
