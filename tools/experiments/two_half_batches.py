"""Experiment: the bench batch (B = 64, T = 60, L = 488) as TWO half batches on two streams of one GPU, each a shard with its own
handle (first_sample = 0 / 32: the same samples as the whole batch), against the whole batch on one stream.  Idea: each half's
kernels have half the workgroups, so two kernels — one per half — share the CUs and one chain's launch boundaries, cold input
staging and tails overlap the other chain's work.  DHW_ENC_WGS=128 / DHW_CONV_WGS=128 make the tiles of a half the whole batch's.
usage: python tools/experiments/two_half_batches.py [iters]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import dhg_amd  # noqa: E402
from dhg_amd import spec  # noqa: E402

torch.set_num_threads(8)
B, L, Lt, T = 64, 488, 30, 60
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
sd = {k: torch.from_numpy(v) for k, v in spec.synthetic_state_dict(2).items()}
inp = spec.synthetic_inputs(B, L, Lt, seed=5)
tx, sv = torch.from_numpy(inp["text"]).cuda(), torch.from_numpy(inp["style"]).cuda()


def model(b):
    m = dhg_amd.DiffusionModel(2, precision="bf16", max_B=b, max_L=L, max_Lt=Lt).eval()
    m.load_state_dict(sd)
    return m


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


whole = model(B)
ref = dhg_amd.sample(whole, tx, sv, L=L, T=T, seed=7)
ms_whole = timed(lambda: dhg_amd.sample(whole, tx, sv, L=L, T=T, seed=7))

os.environ["DHW_ENC_WGS"] = os.environ.get("HALF_ENC_WGS", "128")
os.environ["DHW_CONV_WGS"] = os.environ.get("HALF_CONV_WGS", "128")
halves = [model(B // 2), model(B // 2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
parts = [(tx[:32].contiguous(), sv[:32].contiguous()), (tx[32:].contiguous(), sv[32:].contiguous())]
outs = [None, None]


def both():
    for i in (0, 1):
        with torch.cuda.stream(streams[i]):
            outs[i] = dhg_amd.sample(halves[i], parts[i][0], parts[i][1], L=L, T=T, seed=7, first_sample=32 * i)


def serial():
    for i in (0, 1):
        outs[i] = dhg_amd.sample(halves[i], parts[i][0], parts[i][1], L=L, T=T, seed=7, first_sample=32 * i)


both()
torch.cuda.synchronize()
same = torch.equal(torch.cat(outs), ref)
ms_both = timed(both)
ms_serial = timed(serial)
print(f"whole batch, one stream: {ms_whole:.3f} ms | two halves, two streams: {ms_both:.3f} ms | two halves, one stream: {ms_serial:.3f} ms | "
      f"halves == whole batch bit for bit: {same}")
