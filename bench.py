#!/usr/bin/env python3
"""bench.py — stroke-points/sec reverse-sampled (T=60, L=488), BASELINE.json's metric.

A "step" is one complete T=60 reverse-sampling pass of one prompt batch (B=64 prompts per GPU,
L=488, Lt=30, c=(128,192,256), num_layers=2, bf16 denoiser, random-init weights, synthetic
text/style, device-side N(0,1) noise): 60 denoiser calls + 60 scheduler updates.  Inputs are
resident in HBM when the timed region starts.

N>1: one process per GPU.  Either the caller starts the ranks (`python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N`: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the
environment) or `python bench.py --gpus N` starts them itself: the parent process — which never
touches the GPU and never imports torch — spawns N children with that environment, relays rank 0's
JSON line and exits non-zero unless all N ranks finished.  torch.distributed (RCCL) is used only
for the barrier and the max-over-ranks of the wall time; the sampling loop has no collective.
Prompt shards are independent (rank r samples the global prompts [r*B, (r+1)*B)) => weak scaling.

Prints ONE JSON line (see the driver contract) with extra objects:
  roofline     — for the dominant kernel FUNCTION (launch labels grouped as rocprofv3 groups them): algorithmic FLOPs and
                 bytes per launch over the mean launch duration measured with HIP events on the launch stream (library
                 profile mode); `all_stroke_kernels` = the time-weighted figure over every fused stroke-side kernel
  nl4_mode     — the same workload at the class-default depth num_layers=4 (N=1 only)
  train_step   — a short run of the configs[4] training step (N=1 only; `--train` is the full measurement)
  cpu_baseline — the oracle (CPU restatement of the reference, kind "port") timed on this box's host
                 cores on a bounded sample of the same workload (rank 0, N=1 only)
  fp32_mode    — the same workload through the library's fp32 parity mode (N=1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0       # HBM3E spec
PEAK_F32_TFLOPS = 157.3     # f32-input MFMA (v_mfma_f32_16x16x4_f32) = the f32 vector rate (MI355X_MICROARCH.md)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--train", action="store_true",
                    help="time the training step instead (BASELINE configs[4]: forward + loss + backward + all-reduce + Adam; "
                         "batch 32/GPU, L=480, Lt=50, fp32) — a second metric, not the headline line")
    ap.add_argument("--batch", type=int, default=None, help="prompts per GPU (default 64; 32 with --train)")
    ap.add_argument("--L", type=int, default=None, help="stroke length (default 488; 480 with --train)")
    ap.add_argument("--Lt", type=int, default=None, help="text length (default 30; 50 with --train)")
    ap.add_argument("--T", type=int, default=60)
    ap.add_argument("--num-layers", type=int, default=2)
    ap.add_argument("--precision", default=None, choices=["bf16", "fp32"],
                    help="default bf16 (sampling: the reference's compute dtype target) / fp32 with --train (the reference trains in fp32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--no-fp32", action="store_true", help="skip the other-precision throughput figure (fp32_mode; bf16_mode with --train)")
    ap.add_argument("--no-longseq", action="store_true", help="skip the configs[3] figure (longseq_mode: L=1000, T=1000, batch=32) in the default line")
    ap.add_argument("--no-train-step", action="store_true", help="skip the short configs[4] training-step figure (train_step) in the default line")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--streams", type=int, default=0, help="concurrent prompt sub-batches per GPU (0 = library default)")
    # launcher / distributed plumbing rehearsal on CPU (tests/test_bench_launcher_cpu.py): gloo, no GPU, no library
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a box with fewer GPUs than ranks: rank r uses GPU r %% device_count, gloo instead of RCCL "
                         "for the barrier (RCCL refuses two ranks on one device); the line is marked, never a scaling result")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group (RCCL) even for one rank: runs the barrier / max-over-ranks path on a 1-GPU box")
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--stub-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)
    a = ap.parse_args(argv)
    a.precision = a.precision or ("fp32" if a.train else "bf16")
    a.batch = a.batch or (32 if a.train else 64)
    a.L = a.L or (480 if a.train else 488)
    a.Lt = a.Lt or (50 if a.train else 30)
    return a


# ---------------------------------------------------------------------------------------------- launcher (parent)
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch(args, argv) -> int:
    """Start one rank per GPU as child processes and relay rank 0's JSON line.  The parent makes no GPU / HIP /
    torch call (children are fresh interpreters, nothing is re-exec'd after a GPU touch).  Returns the exit code:
    0 only if every rank exited 0 AND rank 0 reported n_gpus == N with N per-rank timings."""
    import tempfile
    n = args.gpus
    port = _free_port()
    procs = []
    failed = []
    with tempfile.TemporaryFile("w+") as out0f:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                          stdout=out0f if r == 0 else subprocess.DEVNULL))
        try:
            # a rank that dies leaves the others blocked in a collective: stop everything as soon as one exits non-zero
            while any(p.poll() is None for p in procs) and not failed:
                failed = [(r, p.returncode) for r, p in enumerate(procs) if p.poll() not in (None, 0)]
                time.sleep(0.1)
            failed = failed or [(r, p.returncode) for r, p in enumerate(procs) if p.poll() != 0]
        finally:
            for p in procs:              # only the exact children started above
                if p.poll() is None:
                    p.kill()
        out0f.seek(0)
        out0 = out0f.read()
    line = next((ln for ln in reversed(out0.splitlines()) if ln.startswith("{")), None)
    if failed or line is None:
        print(f"bench.py: launcher: ranks failed or missing {failed or '(no JSON line from rank 0)'}", file=sys.stderr)
        return 1
    res = json.loads(line)
    if res.get("n_gpus") != n or len(res.get("per_rank_ms", [])) != n:
        print(f"bench.py: launcher: expected {n} ranks, rank 0 reported n_gpus={res.get('n_gpus')}", file=sys.stderr)
        return 1
    print(line, flush=True)
    return 0


# ---------------------------------------------------------------------------------------------- one rank
_JSON_OUT = None   # the rank's real stdout (worker() points fd 1 at stderr so that only the JSON line reaches stdout)


def emit(line: str):
    print(line, file=_JSON_OUT or sys.stdout, flush=True)


def worker(args):
    import faulthandler
    faulthandler.dump_traceback_later(240, repeat=True, file=sys.stderr)   # a stuck run says where
    # ONE JSON line on stdout is the contract; libraries do not know it (RCCL prints its version banner to stdout under
    # NCCL_DEBUG=VERSION, which the GPU boxes export): everything written to fd 1 from here on goes to stderr, the line goes
    # to a private duplicate of the original stdout.
    global _JSON_OUT
    sys.stdout.flush()
    _JSON_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to print a line for the wrong N")
    dist = None
    if args.stub:
        dev = torch.device("cpu")
    else:
        di = (local_rank % max(1, torch.cuda.device_count())) if args.share_gpu else (local_rank if world > 1 else 0)
        torch.cuda.set_device(di)
        dev = torch.device("cuda", di)
    if world > 1 or args.force_dist:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.stub or args.share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    B, L, Lt, T = args.batch, args.L, args.Lt, args.T
    if args.train and not args.stub:
        return train_worker(args, dev, dist, rank, world)
    if args.stub:
        if rank == args.stub_fail_rank:
            raise SystemExit(3)
        model = None

        def one_step(k):
            time.sleep(0.01 * (1 + rank))
            return torch.zeros(1)

        def sync():
            pass
    else:
        import dhg_amd
        from dhg_amd import _lib, spec
        model = dhg_amd.DiffusionModel(args.num_layers, precision=args.precision, max_B=B, max_L=L, max_Lt=Lt).eval()
        model.load_state_dict({k: torch.from_numpy(v) for k, v in spec.synthetic_state_dict(args.num_layers).items()})
        # each rank owns the prompts [rank*B, rank*B+B) of the global batch; noise is keyed by the global index
        inp = spec.synthetic_inputs_range(rank * B, B, L, Lt, seed=1, T=0)
        text = torch.from_numpy(inp["text"]).to(dev)
        style = torch.from_numpy(inp["style"]).to(dev)

        def one_step(k):
            return dhg_amd.sample(model, text, style, L=L, T=T, seed=1000 + k, first_sample=rank * B)

        def sync():
            torch.cuda.synchronize(dev)

        if args.streams:
            os.environ["DHW_STREAMS"] = str(args.streams)
        if args.no_graph:
            one_step(0)
            _lib.lib().dhw_set_graph(model._handle, 0)

    out = None
    for k in range(args.warmup):
        out = one_step(k)
        sync()
        if rank == 0:
            print(f"[bench] warm-up step {k} done", file=sys.stderr, flush=True)

    def barrier():
        sync()
        if dist is not None:
            dist.barrier()
        sync()

    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        out = one_step(args.warmup + k)
    sync()
    dt_own = time.perf_counter() - t0     # this rank's own K steps (diagnostic: per_rank_ms)
    barrier()
    dt = time.perf_counter() - t0
    per_rank = [dt_own]
    if dist is not None:
        cdev = torch.device("cpu") if (args.stub or args.share_gpu) else dev     # gloo reduces host tensors
        tt = torch.tensor([dt], device=cdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        own = [torch.zeros(1, device=cdev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(own, torch.tensor([dt_own], device=cdev, dtype=torch.float64))
        per_rank = [float(x.item()) for x in own]
    assert out is not None and bool(torch.isfinite(out).all()), "non-finite samples"

    points = world * B * L * args.steps
    value = points / dt
    res = {
        "metric": "stroke-points/sec reverse-sampled (T=60, L=488)",
        "value": value,
        "unit": "stroke-points/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.precision,
        "data": "synthetic",
        "config": {"workload": f"configs[1]: batch={B} prompts/GPU, T={T}, L={L}, Lt={Lt}, d_model=128/192/256, "
                               f"num_layers={args.num_layers}, diffusion_mode=new, random-init weights",
                   "global_batch": world * B, "seq_len": L, "parallelism": f"batch-shard x{world} (no collectives)"},
        "point_steps_per_s": value * T,
        "per_rank_ms": [t / args.steps * 1e3 for t in per_rank],
    }
    if args.stub:
        res["data"] = "stub (launcher rehearsal, no GPU work)"
    if args.share_gpu:
        res["data"] = "synthetic; REHEARSAL: ranks share GPUs (not a scaling measurement)"

    if rank == 0:
        print(f"[bench] {value:.4g} stroke-points/s, {dt / args.steps * 1e3:.2f} ms/step", file=sys.stderr, flush=True)
        if model is not None:
            fl, by = model.work(L, Lt)
            res["work_per_sample_call"] = {"flops": fl, "block_boundary_bytes": by}
            res["model_tflops"] = fl * B * T * world * args.steps / dt / 1e12
            if not args.no_kernel_profile:
                res["roofline"], res["kernels"] = kernel_profile(model, one_step, res["ms_per_step"])
                # the attention stage on its own (244 keys at the L/2 level of this workload; 500 keys: longseq_mode below)
                res["roofline"]["by_function"]["attention"] = attention_figure(model, dev, B, L, Lt)
            if world == 1 and not args.no_fp32 and args.precision == "bf16":
                res["fp32_mode"] = other_mode(args, dev, text, style, precision="fp32", num_layers=args.num_layers)
                if args.num_layers != 4:   # the class default depth (model.py:66; SURVEY 8: "report both, primary = 2")
                    res["nl4_mode"] = other_mode(args, dev, text, style, precision="bf16", num_layers=4)
            if world == 1 and not args.no_longseq and args.precision == "bf16":
                res["longseq_mode"] = longseq_mode(args, dev)
            if world == 1 and not args.no_train_step:
                del model
                torch.cuda.empty_cache()
                res["train_step"] = train_step_figure(args, dev)
            if world == 1 and not args.no_cpu_baseline:
                res["cpu_baseline"] = cpu_baseline(args, spec)
        emit(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def train_worker(args, dev, dist, rank, world):
    """--train: K updates of the native training step (dhg_amd.train_model.GraphedTrainStep) on one synthetic batch per
    rank; N > 1 sums the flat gradient buffer across ranks in five bucketed RCCL all-reduces overlapped with the backward's tail
    (train.GradBucketReducer; data parallel, weak scaling)."""
    import numpy as np
    import torch
    from dhg_amd import spec, train, train_model as tm
    torch.set_num_threads(host_cores())   # (the cgroup share, not os.cpu_count(): oversubscribing it made the host the bottleneck)
    B, L, Lt = args.batch, args.L, args.Lt
    sd = spec.synthetic_state_dict(args.num_layers, 128, 192, 256, seed=0)
    model = tm.TrainModel(sd, num_layers=args.num_layers, device=dev, precision=args.precision)
    opt = train.Adam(model.parameters())
    inp = spec.synthetic_inputs_range(rank * B, B, L, Lt, seed=3, T=0)
    g = torch.Generator().manual_seed(3 + rank)
    strokes = torch.randn(B, L, 2, generator=g)
    batch = {"strokes": torch.cat([strokes, (torch.rand(B, L, 1, generator=g) < 0.1).float()], dim=-1),
             "text": torch.from_numpy(inp["text"]), "style": torch.from_numpy(inp["style"])}
    alpha_set = torch.from_numpy(np.load(os.path.join(ROOT, "tests", "golden", "sched.npz"))["alpha"])
    step_fn = tm.GraphedTrainStep(model, opt, B, L, Lt)

    def sync():
        torch.cuda.synchronize(dev)

    def barrier():
        sync()
        if dist is not None:
            dist.barrier()
        sync()

    losses = []
    for k in range(args.warmup):
        losses.append(step_fn(batch, alpha_set, k + 1, graph=not args.no_graph).clone())
        sync()
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        losses.append(step_fn(batch, alpha_set, args.warmup + k + 1, graph=not args.no_graph).clone())
    sync()
    dt_own = time.perf_counter() - t0
    barrier()
    dt = time.perf_counter() - t0
    per_rank = [dt_own]
    if dist is not None:
        cdev = torch.device("cpu") if args.share_gpu else dev
        tt = torch.tensor([dt], device=cdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        own = [torch.zeros(1, device=cdev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(own, torch.tensor([dt_own], device=cdev, dtype=torch.float64))
        per_rank = [float(x.item()) for x in own]
    lv = [float(x[0]) for x in losses]
    assert all(np.isfinite(lv)), "non-finite loss"
    value = world * B * args.steps / dt
    gemm_tflops = model.last_gemm_flops * args.steps / dt_own / 1e12
    res = {"metric": "training samples/sec (forward + loss + backward + grad all-reduce + clip + Adam; L=480)", "value": value,
           "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "bf16 GEMM operands, f32 accumulate / state",
           "data": "synthetic",
           "config": {"workload": f"configs[4]: training step, batch={B}/GPU, L={L}, Lt={Lt}, d_model=128/192/256, num_layers={args.num_layers}, "
                                  "dropout 0.0 + style Dropout(0.3), Adam + Noam + clip 100, random-init weights",
                      "global_batch": world * B, "seq_len": L,
                      "parallelism": f"data-parallel x{world}" + (" (gradient all-reduce in 5 buckets, each issued behind its graph segment of the backward; Adam applies 1 / world)" if world > 1 else "")},
           "per_rank_ms": [t / args.steps * 1e3 for t in per_rank], "loss_first_last": [lv[0], lv[-1]],
           "launches_per_update": model.last_launches, "launch": "eager" if args.no_graph else "hipGraph",
           "roofline": {"bound": "mfma", "achieved": gemm_tflops, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s", "frac": gemm_tflops / PEAK_F32_TFLOPS,
                        "traffic": None, "kernel": "whole update (all GEMM FLOPs / wall time of the update)",
                        "gemm_flops_per_update": model.last_gemm_flops}}
    if args.share_gpu:
        res["data"] = "synthetic; REHEARSAL: ranks share GPUs (not a scaling measurement)"
    if dist is not None and world == 1 and step_fn._reducer is not None:
        # --force-dist on one GPU: the timed run above went through the process-group path (five graph segments, Adam behind the
        # last bucket) with the one-rank collectives SKIPPED — a SUM over one rank is the buffer itself, which is what the product
        # does.  The same updates again with the five RCCL all-reduces forced, to rehearse the collective path: RCCL runs a one-rank
        # all-reduce as a copy kernel (0.28 ms for the 40 MB buffer), so this figure is RCCL's own cost, not the overlap's.
        step_fn._reducer.reduce_one_rank = True
        for k in range(2):
            step_fn(batch, alpha_set, args.warmup + args.steps + k + 1)
        sync()
        t1 = time.perf_counter()
        for k in range(args.steps):
            step_fn(batch, alpha_set, args.warmup + args.steps + k + 3)
        sync()
        res["rccl_one_rank_forced"] = {"ms_per_step": (time.perf_counter() - t1) / args.steps * 1e3, "steps": args.steps,
                                       "note": "five bucketed RCCL all-reduces issued on a one-rank group (identity; skipped in the timed run)"}
        step_fn._reducer.reduce_one_rank = False
    if rank == 0:
        if world == 1 and args.precision == "fp32" and not args.no_fp32:
            # the same update with mixed-precision GEMMs (bf16-rounded operands, fp32 accumulation, fp32 master weights)
            m2 = tm.TrainModel(sd, num_layers=args.num_layers, device=dev, precision="bf16")
            s2 = tm.GraphedTrainStep(m2, train.Adam(m2.parameters()), B, L, Lt)
            for k in range(2):
                s2(batch, alpha_set, k + 1)
            sync()
            t1 = time.perf_counter()
            for k in range(args.steps):
                s2(batch, alpha_set, k + 3)
            sync()
            d2 = (time.perf_counter() - t1) / args.steps
            res["bf16_mode"] = {"value": B / d2, "unit": "samples/s", "ms_per_step": d2 * 1e3, "steps": args.steps}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_train_baseline(args, spec)
        emit(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_train_baseline(args, spec):
    """The oracle's forward (oracle/ref_cpu.py, the CPU restatement of the reference) under torch autograd with the reference's
    loss — forward + backward of a small batch at the same L on this box's host cores (the optimizer is left out: it is
    <1 % of a CPU step).  kind "port"."""
    import torch
    import torch.nn.functional as F
    from oracle import ref_cpu
    B, L, Lt = 16, args.L, args.Lt
    cores = torch.get_num_threads()
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in spec.synthetic_state_dict(args.num_layers).items()}
    inp = spec.synthetic_inputs_range(0, B, L, Lt, seed=3, T=0)
    g = torch.Generator().manual_seed(1)
    x, eps = torch.randn(B, L, 2, generator=g), torch.randn(B, L, 2, generator=g)
    pen = (torch.rand(B, L, generator=g) < 0.1).float()
    alphas = torch.rand(B, 1, generator=g) * 0.9 + 0.05

    def step():
        xp = torch.sqrt(alphas).unsqueeze(-1) * x + torch.sqrt(1 - alphas).unsqueeze(-1) * eps
        out = ref_cpu.forward(sd, xp, torch.from_numpy(inp["text"]), torch.sqrt(alphas), torch.from_numpy(inp["style"]))
        score, pp = out[0], out[1]
        loss = ((eps - score) ** 2).sum(-1).mean() + (F.binary_cross_entropy(pp, pen.clamp(1e-7, 1 - 1e-7), reduction="none").mean(-1) * alphas.squeeze(-1)).mean()
        loss.backward()
        for v in sd.values():
            v.grad = None
    step()
    best = 1e30
    for _ in range(3):
        t0 = time.perf_counter()
        step()
        best = min(best, time.perf_counter() - t0)
    return {"value": B / best, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"forward + loss + backward of {B} samples (L={L}, Lt={Lt}) through the CPU oracle under torch autograd, best of 3"}


KERNEL_FUNCTION = {"enc.fused_bc": "enc_bc_kernel", "enc.fused_bc+a": "enc_bc_kernel", "enc.fused_a": "enc_a_kernel",
                   "convblock.fused": "convblock_kernel"}   # library launch label -> __global__ function (what rocprofv3 reports)


def kernel_profile(model, one_step, ms_per_step):
    """One un-timed pass with every launch bracketed by HIP events on the launch stream.  Launch labels are grouped by
    kernel FUNCTION (as the rocprofv3 kernel trace groups them): `roofline` is for the function with the largest share of the
    GPU time; `roofline["all_stroke_kernels"]` is the time-weighted figure over every fused stroke-side kernel.

    The profile pass launches eagerly (a replayed hipGraph has no per-kernel events): its event brackets include the gaps the
    timed graph replay does not have, so their sum exceeds the timed step (r3: 21.9 vs 19.7 ms).  The excess is a roughly CONSTANT
    cost per launch (event record + eager dispatch), not a share of the kernel's time, so it is removed as one constant per launch,
    gap = (sum of event times - timed step) / launches — round 4 scaled every time by one factor instead, which under-timed the long
    dominant kernel by ~5 % against rocprofv3 (33.4 vs 35.0 us) and overstated its roofline fraction.  The per-kernel figures then
    add up to the timed step; they are DERIVED figures: the raw event mean of the dominant function is reported beside them
    (`avg_launch_us_raw_event`) and the committed rocprofv3 summary of the same command (profiles/) is the cross-check."""
    import torch
    model.profile(True)
    one_step(10_000)
    torch.cuda.synchronize()
    rows = model.profile_results()
    model.profile(False)
    rows = [r for r in rows if r["launches"]]
    raw_total = sum(r["total_ms"] for r in rows)
    n_launch = sum(r["launches"] for r in rows)
    gap_ms = max(0.0, (raw_total - ms_per_step) / n_launch) if n_launch else 0.0
    # (a kernel shorter than the gap keeps 10 % of its raw time rather than going negative; the remainder is spread again below)
    for r in rows:
        r["raw_ms"] = r["total_ms"]
        r["total_ms"] = max(0.1 * r["total_ms"], r["total_ms"] - gap_ms * r["launches"])
    total = sum(r["total_ms"] for r in rows)
    scale = min(1.0, ms_per_step / total) if total > 0 and gap_ms > 0 else 1.0   # residual after the clamps (within a fraction of a percent of 1)
    for r in rows:
        r["total_ms"] *= scale
    total *= scale
    table = []
    for r in sorted(rows, key=lambda r: -r["total_ms"]):
        us = r["total_ms"] * 1e3 / r["launches"]
        fl, by = r["flops"] / r["launches"], r["bytes"] / r["launches"]
        table.append({"label": r["label"], "function": KERNEL_FUNCTION.get(r["label"], r["label"]), "launches": r["launches"],
                      "share": r["total_ms"] / total, "avg_us": us,
                      "tflops": fl / us / 1e6 if us else 0.0, "gbs": by / us / 1e3 if us else 0.0})
    groups = {}
    for r in rows:
        g = groups.setdefault(KERNEL_FUNCTION.get(r["label"], r["label"]), {"total_ms": 0.0, "raw_ms": 0.0, "launches": 0, "flops": 0.0, "bytes": 0.0, "labels": []})
        for k in ("total_ms", "raw_ms", "launches", "flops", "bytes"):
            g[k] += r[k]
        g["labels"].append(r["label"])
    name, dom = max(((k, g) for k, g in groups.items() if g["flops"] > 0), key=lambda kg: kg[1]["total_ms"])
    us = dom["total_ms"] * 1e3 / dom["launches"]
    fl, by = dom["flops"] / dom["launches"], dom["bytes"] / dom["launches"]
    t_mfma, t_hbm = fl / (PEAK_BF16_TFLOPS * 1e6), by / (PEAK_HBM_GBS * 1e3)   # us at the two roofs
    if t_hbm >= t_mfma:
        roof = {"bound": "hbm", "achieved": by / us / 1e3, "peak": PEAK_HBM_GBS, "unit": "GB/s"}
    else:
        roof = {"bound": "mfma", "achieved": fl / us / 1e6, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s"}
    roof["frac"] = roof["achieved"] / roof["peak"]
    # both roofs north_star names, whichever binds: MFMA fraction and HBM fraction (algorithmic bytes / time / 8 TB/s)
    roof["mfma_frac"] = fl / us / 1e6 / PEAK_BF16_TFLOPS
    roof["hbm_frac"] = by / us / 1e3 / PEAK_HBM_GBS
    stroke = [g for k, g in groups.items() if k in set(KERNEL_FUNCTION.values())]
    s_ms, s_fl, s_by = (sum(g[k] for g in stroke) for k in ("total_ms", "flops", "bytes"))
    traffic = pmc_traffic(name)
    if traffic:
        traffic["algorithmic_bytes_per_launch"] = by
        traffic["ratio"] = traffic["bytes_per_launch"] / by if by else None   # PMC bytes / algorithmic bytes (> 1 = re-reads, padding, hand-over buffers)
    roof.update({"traffic": traffic, "kernel": name, "labels": sorted(dom["labels"]), "avg_launch_us": us, "launches": dom["launches"],
                 "flops_per_launch": fl, "bytes_per_launch": by, "mfma_tflops": fl / us / 1e6, "hbm_gbs": by / us / 1e3,
                 "share_of_gpu_time": dom["total_ms"] / total, "sum_kernel_ms_per_step": total,
                 "eager_event_sum_ms_per_step": raw_total, "eager_gap_us_per_launch": gap_ms * 1e3, "launches_per_step": n_launch, "event_time_scale": scale,
                 "avg_launch_us_raw_event": dom["raw_ms"] * 1e3 / dom["launches"],
                 "timing": "HIP events around eager launches minus one constant per launch (eager_gap_us_per_launch = (event sum - timed graph-replay step) / launches), "
                           "so that the per-kernel times add up to the timed step; derived — cross-check: profiles/ rocprofv3 kernel stats of the same command",
                 "by_function": {k: {"share": g["total_ms"] / total, "avg_us": g["total_ms"] * 1e3 / g["launches"], "launches": g["launches"],
                                     "mfma_frac": g["flops"] / (g["total_ms"] * 1e9) / PEAK_BF16_TFLOPS if g["total_ms"] else 0.0}
                                 for k, g in sorted(groups.items(), key=lambda kg: -kg[1]["total_ms"]) if g["flops"] > 0},
                 "all_stroke_kernels": {"share_of_gpu_time": s_ms / total, "mfma_tflops": s_fl / (s_ms * 1e9) if s_ms else 0.0,
                                        "mfma_frac": s_fl / (s_ms * 1e9) / PEAK_BF16_TFLOPS if s_ms else 0.0,
                                        "hbm_gbs": s_by / (s_ms * 1e6) if s_ms else 0.0}})
    return roof, table


# csrc files that are NOT part of the sampling path (training step, StyleExtractor): left out of the sampler's source hash.
# Every other file under csrc/ is hashed, so a new kernel header can never be silently missing
# (tests/test_host_cpu.py::test_kernel_source_hash_covers_csrc keeps the two lists exhaustive).
NON_SAMPLER_SOURCES = {"train.hip", "dhw_train_api.cpp", "style.hip", "dhw_style_api.cpp", "abi_guard.h"}
SAMPLER_SOURCES = {"convblock.hip", "convblock_core.h", "enclayer.hip", "enc_a_core.h", "enc_bc_core.h", "persist.hip", "persist.h", "gemm.hip", "gemm_core.h",
                   "attn.hip", "attn_core.h", "misc.hip", "textside.hip", "epilogue.h", "heads_core.h", "dhw_common.h", "dhw_kernels.h", "xcd_swizzle.h",
                   "dhw_api.cpp"}


def kernel_source_hash() -> str:
    """Hash of the sampling path's kernel sources (the GPU box has no .git, so a commit id is not available there): the PMC
    file records the hash it was taken at, and a mismatch marks the traffic figure stale.  Everything under csrc/ except the
    training / StyleExtractor translation units (NON_SAMPLER_SOURCES) goes in."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "diffusion-handwriting-generation.pytorch_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f not in NON_SAMPLER_SOURCES and os.path.isfile(os.path.join(d, f)):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(label):
    """HBM-side bytes per launch of the dominant kernel class from the newest committed rocprofv3 PMC passes
    (profiles/r*_hbm_traffic_pmc.json: FETCH_SIZE and WRITE_SIZE collected in separate runs of this same command,
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  PMC collection serialises kernels, so it is
    not repeated inside the timed run.  `stale` = the kernel sources changed since the passes were taken."""
    import glob
    name = KERNEL_FUNCTION.get(label, label if label in KERNEL_FUNCTION.values() else None)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic_pmc.json")))
    if not name or not files:
        return None
    try:
        with open(files[-1]) as f:
            j = json.load(f)
        k = j["kernels"].get(name)
        if not k:
            return None
        out = {"bytes_per_launch": k["hbm_bytes_per_launch"], "fetch": k["fetch_bytes_per_launch"],
               "write": k["write_bytes_per_launch"], "source": "profiles/" + os.path.basename(files[-1]),
               "stale": j.get("kernel_source_hash") != kernel_source_hash()}
        # the same passes per launch POSITION of the denoiser call against activations once + the launch's weights once per XCD
        # (8 private L2s: tools/pmc_traffic_by_dispatch.py) — what the fabric must deliver whatever the kernel does
        by = files[-1].replace("_hbm_traffic_pmc.json", "_hbm_traffic_by_launch.json")
        if os.path.exists(by):
            with open(by) as f:
                t = json.load(f)["kernels"].get(name)
            if t:
                sm = t["sum_over_a_call"]
                out["per_call"] = {"pmc_bytes": sm["pmc_bytes"], "algorithmic_bytes_weights_once": sm["algorithmic_bytes_weights_once"],
                                   "expected_bytes_weights_once_per_xcd": sm["predicted_bytes_weights_per_xcd"],
                                   "ratio_vs_algorithmic": sm["ratio_vs_algorithmic"], "ratio_vs_expected": sm["ratio_vs_predicted"],
                                   "source": "profiles/" + os.path.basename(by)}
        return out
    except (OSError, ValueError, KeyError):
        return None


def other_mode(args, dev, text, style, precision, num_layers):
    """The same workload in another mode of the library — fp32_mode: the fp32 parity mode (exact-f32 MFMA, the kernel set the
    goldens pin at 2e-5); nl4_mode: the class-default depth num_layers=4 in bf16.  One warm-up + 2 timed steps."""
    import torch
    import dhg_amd
    from dhg_amd import spec
    B, L, Lt, T = args.batch, args.L, args.Lt, args.T
    m = dhg_amd.DiffusionModel(num_layers, precision=precision, max_B=B, max_L=L, max_Lt=Lt).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in spec.synthetic_state_dict(num_layers).items()})
    dhg_amd.sample(m, text, style, L=L, T=T, seed=1)
    torch.cuda.synchronize(dev)
    n = 2 if precision == "fp32" else 5
    t0 = time.perf_counter()
    for k in range(n):
        dhg_amd.sample(m, text, style, L=L, T=T, seed=2 + k)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / n
    fl, _ = m.work(L, Lt)
    del m
    return {"value": B * L / dt, "unit": "stroke-points/s", "ms_per_step": dt * 1e3, "steps": n, "num_layers": num_layers, "precision": precision,
            "model_tflops": fl * B * T / dt / 1e12}


def attention_figure(m, dev, B, L, Lt):
    """north_star: "MFMA utilisation for attention against the chip's peak".  One denoiser call at (B, L, Lt), then the
    self-attention stage (QK^T + softmax + PV + K/V staging inside enc_bc_kernel) of every stroke-side EncoderLayer timed on its
    own by the library (dhw_debug_attention_time: the product kernel launched with and without the stage, HIP events)."""
    import torch
    from dhg_amd import spec
    inp = spec.synthetic_inputs(B, L, Lt, seed=2, T=0)
    with torch.no_grad():
        m(torch.from_numpy(inp["strokes"]).to(dev), torch.from_numpy(inp["text"]).to(dev), torch.full((B, 1), 0.7, device=dev),
          torch.from_numpy(inp["style"]).to(dev))
    torch.cuda.synchronize(dev)
    out = {}
    for layer, (name, d, lk) in enumerate([("enc3", 192, L // 2), ("enc5", 256, L // 4)] + [(f"att_layers.{i}", 384, L // 8) for i in range(m.num_layers)]):
        w, wo, fl = m.attention_time(layer, 20)
        us = max(w - wo, 1e-3)
        out[name] = {"keys": lk, "d": d, "heads": d // 64, "us": us, "enc_bc_us": w, "enc_bc_us_without_stage": wo, "flops": fl,
                     "mfma_tflops": fl / us / 1e6, "mfma_frac": fl / us / 1e6 / PEAK_BF16_TFLOPS}
    return out


def longseq_mode(args, dev):
    """BASELINE configs[3]: L=1000 strokes, T=1000 steps, batch=32 per GPU, Lt=62, bf16 (output_dense x 0.05 as in
    tests/test_gpu_round2.py: a random-init free-running reverse process otherwise leaves fp32 range by step ~900, in the
    reference's arithmetic too).  One warm-up (graph capture of the 1000-step loop) + 2 timed 1000-step batches; plus the
    attention stage at this length (500 keys at the L/2 level)."""
    import torch
    import dhg_amd
    from dhg_amd import spec
    B, L, Lt, T = 32, 1000, 62, 1000
    sd = {k: torch.from_numpy(v) for k, v in spec.synthetic_state_dict(args.num_layers).items()}
    sd["output_dense.weight"] = sd["output_dense.weight"] * 0.05
    sd["output_dense.bias"] = sd["output_dense.bias"] * 0.05
    m = dhg_amd.DiffusionModel(args.num_layers, precision="bf16", max_B=B, max_L=L, max_Lt=Lt).eval()
    m.load_state_dict(sd)
    inp = spec.synthetic_inputs(B, L, Lt, seed=5, T=0)
    tx, sv = torch.from_numpy(inp["text"]).to(dev), torch.from_numpy(inp["style"]).to(dev)
    out = dhg_amd.sample(m, tx, sv, L=L, T=T, seed=3)
    torch.cuda.synchronize(dev)
    n = 2
    t0 = time.perf_counter()
    for k in range(n):
        out = dhg_amd.sample(m, tx, sv, L=L, T=T, seed=4 + k)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / n
    assert bool(torch.isfinite(out).all()), "non-finite samples (configs[3])"
    fl, _ = m.work(L, Lt)
    res = {"workload": f"configs[3]: batch={B}, T={T}, L={L}, Lt={Lt}, bf16, num_layers={args.num_layers}, output_dense x 0.05",
           "value": B * L / dt, "unit": "stroke-points/s", "ms_per_step": dt * 1e3, "ms_per_denoiser_call": dt * 1e3 / T, "steps": n,
           "model_tflops": fl * B * T / dt / 1e12, "model_mfma_frac": fl * B * T / dt / 1e12 / PEAK_BF16_TFLOPS,
           "attention": attention_figure(m, dev, B, L, Lt)}
    del m
    return res


def train_step_figure(args, dev):
    """BASELINE configs[4] on this GPU: a few updates of the native training step (forward + loss + backward + clip + Adam;
    batch 32, L=480, Lt=50, fp32, hipGraph replay) — `python bench.py --train` is the full form of this measurement."""
    import numpy as np
    import torch
    from dhg_amd import spec, train, train_model as tm
    torch.set_num_threads(host_cores())
    B, L, Lt = 32, 480, 50
    sd = spec.synthetic_state_dict(args.num_layers, 128, 192, 256, seed=0)
    model = tm.TrainModel(sd, num_layers=args.num_layers, device=dev, precision="fp32")
    opt = train.Adam(model.parameters())
    inp = spec.synthetic_inputs_range(0, B, L, Lt, seed=3, T=0)
    g = torch.Generator().manual_seed(3)
    strokes = torch.randn(B, L, 2, generator=g)
    batch = {"strokes": torch.cat([strokes, (torch.rand(B, L, 1, generator=g) < 0.1).float()], dim=-1),
             "text": torch.from_numpy(inp["text"]), "style": torch.from_numpy(inp["style"])}
    alpha_set = torch.from_numpy(np.load(os.path.join(ROOT, "tests", "golden", "sched.npz"))["alpha"])
    step_fn = tm.GraphedTrainStep(model, opt, B, L, Lt)
    losses = [float(step_fn(batch, alpha_set, k + 1)[0]) for k in range(3)]
    torch.cuda.synchronize(dev)
    n = 10
    t0 = time.perf_counter()
    for k in range(n):
        step_fn(batch, alpha_set, 4 + k)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / n
    gemm_tflops = model.last_gemm_flops / dt / 1e12
    assert all(np.isfinite(losses)), "non-finite training loss"
    return {"ms_per_update": dt * 1e3, "samples_per_s": B / dt, "batch": B, "L": L, "Lt": Lt, "dtype": "f32", "updates": n,
            "launches_per_update": model.last_launches, "gemm_tflops": gemm_tflops, "gemm_roofline_frac": gemm_tflops / PEAK_F32_TFLOPS,
            "peak_f32_mfma_tflops": PEAK_F32_TFLOPS, "workload": "configs[4] per-GPU shard: forward + diffusion loss + backward + clip + Adam, hipGraph replay"}


def host_cores() -> int:
    """CPU cores this process may really use: affinity mask, capped by the cgroup CPU quota (a GPU box
    exposes every host core in the mask but grants a 16-core share per GPU; oversubscribing it stalls)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("DHW_CPU_THREADS", "16"))))


def cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, spec):
    """The oracle on this box's host cores: full T-step loop with autograd recording on as in the
    reference (inference.py:84-94), B=1 (BASELINE config #1) and B=8, best of 3 after one warm-up each, plus the
    no_grad variant at B=1.  `value` is the B=1 figure (the reference's own configuration)."""
    import torch
    from oracle import ref_cpu
    threads = host_cores()
    torch.set_num_threads(threads)
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in spec.synthetic_state_dict(args.num_layers).items()}

    def best(B, grad, n=3):
        inp = spec.synthetic_inputs(B, args.L, args.Lt, seed=1, T=args.T)
        text, style, noise = (torch.from_numpy(inp[k]) for k in ("text", "style", "noise"))
        ts = []
        for i in range(n + 1):    # run 0 = warm-up
            t0 = time.perf_counter()
            ref_cpu.sample(sd, text, style, args.L, noise, T=args.T, grad=grad)
            ts.append(time.perf_counter() - t0)
        return min(ts[1:]), ts

    t1, all1 = best(1, True)
    print(f"[bench] cpu_baseline B=1: {t1:.2f}s per prompt on {threads} threads", file=sys.stderr, flush=True)
    t1n, _ = best(1, False, n=2)
    t8, all8 = best(8, True, n=2 if all1[0] > 2.0 else 3)
    return {"value": args.L / t1, "unit": "stroke-points/s", "cores": threads, "kind": "port", "cpu": cpu_model(),
            "sample": f"B=1 prompt, T={args.T}, L={args.L}, Lt={args.Lt}: full {args.T}-step loop, autograd recording on "
                      f"as in the reference; best of 3 after 1 warm-up ({t1:.2f} s per prompt)",
            "runs_s": [round(t, 3) for t in all1],
            "no_grad_value": args.L / t1n,
            "b8_value": 8 * args.L / t8, "b8_runs_s": [round(t, 3) for t in all8]}


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch(args, argv))
    worker(args)


if __name__ == "__main__":
    main()
