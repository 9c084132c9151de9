#!/usr/bin/env python3
"""Where the persistent step kernel's time goes: per phase, over all workgroups of the last step of a sample() call, the time from
the ticket to the inputs being ready (wait on the previous phase of the tile's sample), the body, and the span of the phase.
Usage (GPU box): DHW_PERSIST_TRACE=1 python tools/persist_trace.py [B=64] [L=488] [T=6]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DHW_PERSIST_TRACE", "1")
os.environ.setdefault("DHW_PERSIST", "1")
import dhg_amd  # noqa: E402
from dhg_amd import _lib, spec  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = int(sys.argv[2]) if len(sys.argv) > 2 else 488
T = int(sys.argv[3]) if len(sys.argv) > 3 else 6
Lt = 30
m = dhg_amd.DiffusionModel(2, precision="bf16", max_B=B, max_L=L, max_Lt=Lt).eval()
m.load_state_dict({k: torch.from_numpy(v) for k, v in spec.synthetic_state_dict(2).items()})
inp = spec.synthetic_inputs(B, L, Lt, seed=12, T=T)
tx, sv = (torch.from_numpy(inp[k]).cuda() for k in ("text", "style"))
for _ in range(3):
    out = dhg_amd.sample(m, tx, sv, L=L, T=T, seed=1)
torch.cuda.synchronize()
NP = 16
buf = np.zeros(1024 * NP * 4, dtype=np.uint64)
n = _lib.lib().dhw_debug_persist_trace(m._handle, buf.ctypes.data_as(C.c_void_p), buf.size)
assert n > 0, "no trace (DHW_PERSIST_TRACE=1?)"
tr = buf[: n * NP * 4].reshape(n, NP, 4).astype(np.int64)
t_in = tr[:, 0, 3]
t_out = tr[:, 1, 3] & ((1 << 56) - 1)
xcc = (tr[:, 1, 3] >> 56) & 0xff
t0 = t_in.min()
us = lambda v: (v - t0) / 100.0
print(f"B={B} L={L}: {n} workgroups; workgroups per XCC id: {np.bincount(xcc, minlength=8).tolist()}")
print(f"kernel entry spread {us(t_in.max()):.2f} us; exit first {us(t_out.min()):.2f} last {us(t_out.max()):.2f} us")
names = ["enc1", "enc2+3a", "enc3bc", "enc4", "enc5a", "enc5bc+att0a", "att0bc+att1a", "att1bc", "dec3", "dec2", "dec1"]
print(f"{'phase':14s} {'tiles':>5s} {'ticket(first/med/last)':>26s} {'wait med/max':>14s} {'body med/max':>14s} {'done(first/med/last)':>24s}")
for ph in range(NP):
    act = tr[:, ph, 1] > 0          # workgroups that ran a tile in this phase (inputs-ready stamp)
    if not act.any():
        continue
    tk, rd, dn = tr[act, ph, 0], tr[act, ph, 1], tr[act, ph, 2]
    wait, body = (rd - tk) / 100.0, (dn - rd) / 100.0
    nm = names[ph] if ph < len(names) else str(ph)
    print(f"{nm:14s} {int(act.sum()):5d} {us(tk.min()):8.2f}/{us(np.median(tk)):7.2f}/{us(tk.max()):7.2f} {np.median(wait):6.2f}/{wait.max():6.2f} "
          f"{np.median(body):6.2f}/{body.max():6.2f} {us(dn.min()):8.2f}/{us(np.median(dn)):7.2f}/{us(dn.max()):7.2f}")
