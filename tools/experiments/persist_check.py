# quick GPU check of the persistent step kernel: samples of the persistent form vs the per-kernel form, several shapes
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import dhg_amd
from dhg_amd import spec
def model(B, L, Lt, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        m = dhg_amd.DiffusionModel(2, precision="bf16", max_B=B, max_L=L, max_Lt=Lt).eval()
        m.load_state_dict({k: torch.from_numpy(v) for k, v in spec.synthetic_state_dict(2).items()})
        # force handle creation under this env
        return m
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
for (B, L, Lt, T) in ((3, 80, 9, 4), (16, 488, 30, 6), (64, 488, 30, 6)):
    inp = spec.synthetic_inputs(B, L, Lt, seed=12, pad=1, T=T)
    tx, sv, nz = (torch.from_numpy(inp[k]).cuda() for k in ("text", "style", "noise"))
    outs = {}
    for name, env in (("persist", {"DHW_PERSIST": "1"}), ("kernels", {"DHW_PERSIST": "0"})):
        os.environ.update(env) if env else os.environ.pop("DHW_PERSIST", None)
        m = dhg_amd.DiffusionModel(2, precision="bf16", max_B=B, max_L=L, max_Lt=Lt).eval()
        m.load_state_dict({k: torch.from_numpy(v) for k, v in spec.synthetic_state_dict(2).items()})
        o = dhg_amd.sample(m, tx, sv, L=L, T=T, noise=nz)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(3): o = dhg_amd.sample(m, tx, sv, L=L, T=T, noise=nz)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / 3
        outs[name] = o.cpu()
        print(f"B={B} L={L} T={T} {name}: plans={m.persistent_plans()} {dt*1e3/T:.3f} ms/step finite={bool(torch.isfinite(o).all())}", flush=True)
        os.environ.pop("DHW_PERSIST", None)
        del m
    d = (outs["persist"] - outs["kernels"]).abs().max().item()
    print(f"   max |persist - kernels| = {d:.3e}  equal={torch.equal(outs['persist'], outs['kernels'])}  scale={outs['kernels'].abs().max().item():.2f}", flush=True)
