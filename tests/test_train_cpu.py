"""Host logic of the training step's first slice, pinned by fixtures the imported reference generated
(oracle/make_golden_r2.py): get_alphas' RNG call order, the Noam schedule, and the world-size-2 gradient all-reduce (gloo)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import dhg_amd
from dhg_amd import train


def test_get_alphas_draws_what_the_reference_draws(golden_dir):
    g = np.load(os.path.join(golden_dir, "train_sched.npz"))
    alpha_set = dhg_amd.get_alpha_set()
    torch.manual_seed(123)
    a = train.get_alphas(16, alpha_set)
    assert a.shape == (16, 1)
    assert np.allclose(a.numpy(), g["alphas"], rtol=0, atol=1e-7)          # same randint / rand draws, same interpolation
    lo, hi = alpha_set[torch.from_numpy(g["idx"])], alpha_set[torch.from_numpy(g["idx"]) + 1]
    assert torch.all((a <= lo) & (a >= hi))                                # inside its schedule interval (abar decreases)


def test_noam_schedule_matches_the_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "train_sched.npz"))
    for warm, key in ((4, "lr_warm4"), (10000, "lr_warm10000")):
        got = np.array([train.noam_lr(s, 256, warm) for s in range(1, 21)])
        assert np.allclose(got, g[key], rtol=1e-12, atol=0), warm


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(5)
    full = [torch.randn(world, 1000, generator=g), torch.randn(world, 37, generator=g)]      # per-rank gradients, known to all
    mine = [f[rank].clone() for f in full]
    train.allreduce_grads(mine)
    ret[rank] = all(torch.allclose(m, f.mean(0), atol=1e-6) for m, f in zip(mine, full))
    dist.destroy_process_group()


def test_gradient_allreduce_averages_over_ranks_gloo():
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        assert dict(ret) == {0: True, 1: True}
