#!/usr/bin/env python3
"""Per-GEMM shapes and (host-bound: eager ctypes launches take longer than the kernels, use gemm_time.py or tools/bench_sgemm for
kernel times) timing of one training update (configs[4] per-GPU shard: batch 32, L=480, Lt=50, fp32), eager launches with a pair
of events around each dhw_op_gemm: which shapes the 73 % of the update spent in GEMMs go to.  Diagnostic, GPU only."""
import collections
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dhg_amd  # noqa: E402
from dhg_amd import spec, train, train_model as tm  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B, L, Lt = 32, 480, 50
    sd = spec.synthetic_state_dict(2, 128, 192, 256, seed=0)
    model = tm.TrainModel(sd, num_layers=2, device=dev, precision=os.environ.get("PREC", "fp32"))
    opt = train.Adam(model.parameters())
    inp = spec.synthetic_inputs_range(0, B, L, Lt, seed=3, T=0)
    g = torch.Generator().manual_seed(3)
    strokes = torch.randn(B, L, 2, generator=g)
    batch = {"strokes": torch.cat([strokes, (torch.rand(B, L, 1, generator=g) < 0.1).float()], dim=-1),
             "text": torch.from_numpy(inp["text"]), "style": torch.from_numpy(inp["style"])}
    alpha_set = torch.from_numpy(np.load(os.path.join(ROOT, "tests", "golden", "sched.npz"))["alpha"])
    step = tm.GraphedTrainStep(model, opt, B, L, Lt)
    step(batch, alpha_set, 1, graph=False)
    torch.cuda.synchronize()
    orig = tm.Tape.gemm
    log = []

    def timed(self, A, a_off, sam, sak, Bm, b_off, sbk, sbn, Cm, c_off, scm, scn, M, N, K, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st = torch.cuda.current_stream()
        e0.record(st)
        orig(self, A, a_off, sam, sak, Bm, b_off, sbk, sbn, Cm, c_off, scm, scn, M, N, K, **kw)
        e1.record(st)
        form = ("AT" if sam == 1 and sak != 1 else "A") + ("BT" if sbk == 1 and sbn != 1 else "B")
        log.append((M, N, K, kw.get("nzo", 1) * kw.get("nzi", 1), kw.get("taps", 1), form, e0, e1))
    tm.Tape.gemm = timed
    step(batch, alpha_set, 2, graph=False)
    torch.cuda.synchronize()
    agg = collections.OrderedDict()
    for M, N, K, Z, taps, form, e0, e1 in log:
        key = (M, N, K, Z, taps, form)
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += e0.elapsed_time(e1) * 1e3
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for _, v in rows)
    print(f"{len(log)} GEMMs, {tot / 1e3:.2f} ms (event-timed, eager: includes launch gaps)")
    print("     M     N     K    Z taps form  calls   us/call   total_us   TFLOP/s  share")
    for (M, N, K, Z, taps, form), (n, us) in rows[:40]:
        fl = 2.0 * M * N * K * Z
        print(f"{M:6d} {N:5d} {K:5d} {Z:4d} {taps:4d} {form:5s} {n:5d} {us / n:9.1f} {us:10.1f} {fl * n / us / 1e6:9.1f} {us / tot:6.3f}")
    json.dump([[list(k), v] for k, v in rows], open(os.path.join(ROOT, "gpurun_out", "train_gemm_shapes.json"), "w"))


if __name__ == "__main__":
    main()
