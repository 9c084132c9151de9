#!/usr/bin/env python3
"""bench.py — stroke-points/sec reverse-sampled (T=60, L=488), BASELINE.json's metric.

A "step" is one complete T=60 reverse-sampling pass of one prompt batch (B=64 prompts per GPU,
L=488, Lt=30, c=(128,192,256), num_layers=2, bf16 denoiser, random-init weights, synthetic
text/style, device-side N(0,1) noise): 60 denoiser calls + 60 scheduler updates.  Inputs are
resident in HBM when the timed region starts.  N>1: one process per GPU (torch.distributed, RCCL
only for the barrier / max-over-ranks of the time; the sampling loop has no collective), prompt
shards are independent => weak scaling.

Prints ONE JSON line (see the driver contract) with two extra objects:
  roofline     — for the dominant kernel class: algorithmic FLOPs and bytes per launch over the mean
                 launch duration measured with HIP events on the launch stream (library profile mode)
  cpu_baseline — the oracle (CPU restatement of the reference, kind "port") timed on this box's host
                 cores on a bounded sample of the same workload (rank 0, N=1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0       # HBM3E spec


def main():
    import faulthandler
    faulthandler.dump_traceback_later(240, repeat=True, file=sys.stderr)   # a stuck run says where
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="prompts per GPU")
    ap.add_argument("--L", type=int, default=488)
    ap.add_argument("--Lt", type=int, default=30)
    ap.add_argument("--T", type=int, default=60)
    ap.add_argument("--num-layers", type=int, default=2)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--streams", type=int, default=0, help="concurrent prompt sub-batches per GPU (0 = library default)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    import dhg_amd
    from dhg_amd import _lib, spec

    B, L, Lt, T = args.batch, args.L, args.Lt, args.T
    model = dhg_amd.DiffusionModel(args.num_layers, precision=args.precision, max_B=B, max_L=L, max_Lt=Lt).eval()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in spec.synthetic_state_dict(args.num_layers).items()})
    # each rank owns the prompts [rank*B, rank*B+B) of the global batch; noise is keyed by the global index
    inp = spec.synthetic_inputs_range(rank * B, B, L, Lt, seed=1, T=0)
    text = torch.from_numpy(inp["text"]).to(dev)
    style = torch.from_numpy(inp["style"]).to(dev)

    def one_step(k):
        return dhg_amd.sample(model, text, style, L=L, T=T, seed=1000 + k, first_sample=rank * B)

    out = None
    if args.streams:
        os.environ["DHW_STREAMS"] = str(args.streams)
    if args.no_graph:
        one_step(0)
        _lib.lib().dhw_set_graph(model._handle, 0)
    for k in range(args.warmup):
        out = one_step(k)
        torch.cuda.synchronize(dev)
        if rank == 0:
            print(f"[bench] warm-up step {k} done", file=sys.stderr, flush=True)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        out = one_step(args.warmup + k)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
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
    }

    if rank == 0:
        print(f"[bench] {value:.4g} stroke-points/s, {dt / args.steps * 1e3:.2f} ms/step", file=sys.stderr, flush=True)
        fl, by = model.work(L, Lt)
        res["work_per_sample_call"] = {"flops": fl, "block_boundary_bytes": by}
        res["model_tflops"] = fl * B * T * world * args.steps / dt / 1e12
        if not args.no_kernel_profile:
            res["roofline"], res["kernels"] = kernel_profile(model, one_step)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args, spec)
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def kernel_profile(model, one_step):
    """One un-timed pass with every launch bracketed by HIP events on the launch stream."""
    model.profile(True)
    one_step(10_000)
    torch.cuda.synchronize()
    rows = model.profile_results()
    model.profile(False)
    rows = [r for r in rows if r["launches"]]
    total = sum(r["total_ms"] for r in rows)
    table = []
    for r in sorted(rows, key=lambda r: -r["total_ms"]):
        us = r["total_ms"] * 1e3 / r["launches"]
        fl, by = r["flops"] / r["launches"], r["bytes"] / r["launches"]
        table.append({"label": r["label"], "launches": r["launches"], "share": r["total_ms"] / total, "avg_us": us,
                      "tflops": fl / us / 1e6 if us else 0.0, "gbs": by / us / 1e3 if us else 0.0})
    dom = next(r for r in sorted(rows, key=lambda r: -r["total_ms"]) if r["flops"] > 0)
    us = dom["total_ms"] * 1e3 / dom["launches"]
    fl, by = dom["flops"] / dom["launches"], dom["bytes"] / dom["launches"]
    t_mfma, t_hbm = fl / (PEAK_BF16_TFLOPS * 1e6), by / (PEAK_HBM_GBS * 1e3)   # us at the two roofs
    if t_hbm >= t_mfma:
        roof = {"bound": "hbm", "achieved": by / us / 1e3, "peak": PEAK_HBM_GBS, "unit": "GB/s"}
    else:
        roof = {"bound": "mfma", "achieved": fl / us / 1e6, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s"}
    roof["frac"] = roof["achieved"] / roof["peak"]
    roof.update({"traffic": pmc_traffic(dom["label"]), "kernel": dom["label"], "avg_launch_us": us, "launches": dom["launches"],
                 "flops_per_launch": fl, "bytes_per_launch": by, "mfma_tflops": fl / us / 1e6, "hbm_gbs": by / us / 1e3,
                 "share_of_gpu_time": dom["total_ms"] / total, "sum_kernel_ms_per_step": total})
    return roof, table


PMC_FILE = "r01_v9_hbm_traffic_pmc.json"


def pmc_traffic(label):
    """HBM-side bytes per launch of the dominant kernel class from the committed rocprofv3 PMC passes
    (FETCH_SIZE and WRITE_SIZE collected in separate runs of this same command, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  PMC collection serialises kernels, so it is not repeated inside
    the timed run; None when no measurement for this kernel class has been committed."""
    name = {"enc.fused_bc": "enc_bc_kernel", "enc.fused_bc+a": "enc_bc_kernel", "enc.fused_a": "enc_a_kernel",
            "convblock.fused": "convblock_kernel"}.get(label)
    path = os.path.join(ROOT, "profiles", PMC_FILE)
    if not name or not os.path.exists(path):
        return None
    try:
        with open(path) as f:
            k = json.load(f)["kernels"].get(name)
        return {"bytes_per_launch": k["hbm_bytes_per_launch"], "fetch": k["fetch_bytes_per_launch"],
                "write": k["write_bytes_per_launch"], "source": "profiles/" + PMC_FILE} if k else None
    except (OSError, ValueError, KeyError):
        return None


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


def cpu_baseline(args, spec):
    """The oracle on this box's host cores: B=1 (BASELINE config #1), full T-step loop, autograd on as in
    the reference (inference.py:84-94), one warm-up + best of 2; plus the no_grad variant."""
    from oracle import ref_cpu
    threads = host_cores()
    torch.set_num_threads(threads)
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in spec.synthetic_state_dict(args.num_layers).items()}
    B = 1
    inp = spec.synthetic_inputs(B, args.L, args.Lt, seed=1, T=args.T)
    text, style, noise = (torch.from_numpy(inp[k]) for k in ("text", "style", "noise"))

    def run(grad):
        t0 = time.perf_counter()
        ref_cpu.sample(sd, text, style, args.L, noise, T=args.T, grad=grad)
        return time.perf_counter() - t0

    t_warm = run(True)
    print(f"[bench] cpu_baseline warm-up {t_warm:.1f}s on {threads} threads", file=sys.stderr, flush=True)
    t_grad = run(True)
    t_nograd = run(False)
    return {"value": B * args.L / t_grad, "unit": "stroke-points/s", "cores": threads, "kind": "port",
            "sample": f"B=1 prompt, T={args.T}, L={args.L}, Lt={args.Lt}: full {args.T}-step loop, autograd recording on "
                      f"as in the reference; 1 timed run after 1 warm-up ({t_grad:.2f} s per prompt)",
            "no_grad_value": B * args.L / t_nograd}


if __name__ == "__main__":
    main()
