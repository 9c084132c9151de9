#!/usr/bin/env python3
"""Where the +0.25 ms of `bench.py --train --force-dist` over the plain run go (one-rank RCCL group: the collectives move nothing).
Variants of the same 32 x 480 update, 30 timed replays each:
  plain        : no process group, ONE graph incl. clip + Adam
  eager_adam   : no process group, one graph WITHOUT the optimizer, clip + Adam as 3 eager launches behind it
  seg_noreduce : the five graph segments of the bucketed path, no collective at all, eager Adam
  seg_bucketed : the product's path under a one-rank "nccl" group (5 segments, 5 async all-reduces, eager Adam)
  flat         : DHW_TRAIN_BUCKETS=0 under the group (one graph, one all-reduce, eager Adam)
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import torch.distributed as dist
from dhg_amd import spec, train, train_model as tm

mode = sys.argv[1]
torch.set_num_threads(8)   # (the host side assembles the batch with small torch CPU ops: uncapped threads on a shared box cost ~15 ms per update)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
if mode in ("seg_bucketed", "flat", "seg_noreduce"):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29581")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
if mode == "flat":
    tm.GRAD_BUCKETS = False
B, L, Lt = 32, 480, 50
sd = spec.synthetic_state_dict(2, 128, 192, 256, seed=0)
model = tm.TrainModel(sd, num_layers=2, device=dev)
opt = train.Adam(model.parameters())
inp = spec.synthetic_inputs_range(0, B, L, Lt, seed=3, T=0)
g = torch.Generator().manual_seed(3)
batch = {"strokes": torch.cat([torch.randn(B, L, 2, generator=g), (torch.rand(B, L, 1, generator=g) < 0.1).float()], dim=-1),
         "text": torch.from_numpy(inp["text"]), "style": torch.from_numpy(inp["style"])}
alpha_set = torch.from_numpy(np.load(os.path.join(ROOT, "tests", "golden", "sched.npz"))["alpha"])
step = tm.GraphedTrainStep(model, opt, B, L, Lt)
if mode == "eager_adam":
    # capture without the optimizer, then apply it eagerly: emulate by pretending a process group exists only for the capture decision
    orig = step.__call__
    def call(batch, alpha_set, k):
        if step.graph is None:
            step._opt_in_graph = False
            step.graph = torch.cuda.CUDAGraph()
            # stage inputs once through the normal path first (warm-up call below did)
            with torch.cuda.graph(step.graph):
                step._body()
        step.graph.replay()
        step._apply(None)
        return step.out
    # one normal eager call to fill the staging buffers
    step(batch, alpha_set, 1, graph=False)
    fn = call
elif mode == "seg_noreduce":
    class Fake:
        def __init__(self): pass
        def launch(self, i): pass
        def wait(self, **k): pass
    step._reducer = Fake()
    fn = lambda b, a, k: step(b, a, k)
else:
    fn = lambda b, a, k: step(b, a, k)
for k in range(3):
    fn(batch, alpha_set, k + 1)
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for k in range(n):
    fn(batch, alpha_set, k + 4)
torch.cuda.synchronize()
print(f"{mode:14s} {(time.perf_counter() - t0) / n * 1e3:.3f} ms per update  (segments: {0 if step.segments is None else len(step.segments)})", flush=True)
if dist.is_initialized():
    dist.destroy_process_group()
