#!/usr/bin/env python3
"""tests/golden/config3_oracle.npz — BASELINE configs[3] (L=1000, Lt=62, T=1000) for TWO prompts through the CPU oracle
(oracle/ref_cpu.py, itself pinned to the imported reference by make_golden.py's fixtures at T=60; the reference hard-codes
T=60 in get_beta_set, utils/nn.py:21, so the T-generalised loop exists only in the oracle).  The oracle needs ~0.3 s per
step at L=1000, too slow to re-run inside every `pytest -m gpu`, hence this committed fixture.

TEACHER FORCED (round 3; SURVEY 7 "hard parts"): the reverse process of a random-init model gains 1/sqrt(1 - beta_i) per
step — 1e15 over this schedule — so a free-running trajectory saturates every probability and makes tolerances meaningless
(round 2's fixture ended at |x| ~ 1e15).  Here the state is recorded and RESET to a fresh seeded N(0,1) draw every EVERY = 16
steps (the largest betas, 0.42, give 1.31^16 = 75 per segment) (oracle: ref_cpu.sample(teacher=...); library: dhw_debug_set_teacher), so |x| stays O(10) while all 1000 schedule
indices, FiLM rows and text-plane chunks are exercised.  Weights are the plain synthetic ones (no head scaling any more).

Inputs are regenerated from seeds by the test: spec.synthetic_inputs(2, 1000, 62, seed=5, T=1000), synthetic_state_dict(2),
resets = torch.randn(62, 2, 1000, 2, generator=manual_seed(1234)).  Stored: the 62 captured states (every 4th stroke row, to keep
the fixture small) and the final [2,1000,3].

    python oracle/make_config3_fixture.py          # ~10 min on 8 cores
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dhg_amd  # noqa: E402,F401
from dhg_amd import spec  # noqa: E402
from oracle import ref_cpu  # noqa: E402

B, L, Lt, T, EVERY, RESET_SEED, ROW_STRIDE = 2, 1000, 62, 1000, 16, 1234, 4


def resets():
    return torch.randn((T - 1) // EVERY, B, L, 2, generator=torch.Generator().manual_seed(RESET_SEED))


def main():
    sd = {k: torch.from_numpy(v) for k, v in spec.synthetic_state_dict(2).items()}
    inp = spec.synthetic_inputs(B, L, Lt, seed=5, T=T)
    tx, sv, nz = (torch.from_numpy(inp[k]) for k in ("text", "style", "noise"))
    t0 = time.time()
    out, snaps = ref_cpu.sample(sd, tx, sv, L, nz, T=T, grad=False, teacher=(EVERY, resets()))
    assert torch.isfinite(out).all()
    caps = torch.stack([snaps[k] for k in sorted(snaps)])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "config3_oracle.npz"), out=out.numpy(), captures=caps[:, :, ::ROW_STRIDE].numpy(), row_stride=ROW_STRIDE,
                        B=B, L=L, Lt=Lt, T=T, seed=5, every=EVERY, reset_seed=RESET_SEED)
    print(f"done in {time.time() - t0:.0f}s; max|x| final {out[..., :2].abs().max().item():.4g}, captures {caps.abs().max().item():.4g}; "
          f"pen in ({out[..., 2].min().item():.3g}, {out[..., 2].max().item():.3g})")


if __name__ == "__main__":
    main()
