import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import dhg_amd
from dhg_amd import spec
from oracle import ref_cpu
import test_gpu_parity as T
for (B, L, Lt, TT) in ((96, 488, 30, 1), (96, 488, 30, 2), (40, 1000, 62, 1), (64, 488, 30, 1), (96, 488, 30, 8)):
    inp = spec.synthetic_inputs(B, L, Lt, seed=100 + B, T=TT)
    tx, sv, nz = (torch.from_numpy(inp[k]).cuda() for k in ("text", "style", "noise"))
    m = dhg_amd.DiffusionModel(2, precision="fp32", max_B=B, max_L=L, max_Lt=Lt).eval()
    m.load_state_dict(T._sd(2))
    out = dhg_amd.sample(m, tx, sv, L=L, T=TT, noise=nz).cpu()
    m1 = dhg_amd.DiffusionModel(2, precision="fp32", max_B=1, max_L=L, max_Lt=Lt).eval()
    m1.load_state_dict(T._sd(2))
    for b in (0, B - 1):
        one = dhg_amd.sample(m1, tx[b:b+1], sv[b:b+1], L=L, T=TT, noise=nz[:, b:b+1]).cpu()
        want, _ = ref_cpu.sample(T._sd(2), tx[b:b+1].cpu(), sv[b:b+1].cpu(), L, nz[:, b:b+1].cpu(), T=TT)
        print(B, L, Lt, TT, "b", b, "batch-vs-single", float((out[b:b+1]-one).abs().max()), "single-vs-oracle", float((one-want).abs().max()), "batch-vs-oracle", float((out[b:b+1]-want).abs().max()))
    del m, m1
