"""N>1 host logic on CPU (gloo, world_size 2): prompt shards are derived from the GLOBAL sample index, so
the union of the ranks' inputs equals the single-process batch, and the bench's max-over-ranks timing
reduction works.  (The sampling loop itself has no collective; the GPU side is covered by -m gpu tests.)"""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import dhg_amd  # noqa: F401
    from dhg_amd import spec
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, L, Lt = 3, 16, 5
    mine = spec.synthetic_inputs_range(rank * B, B, L, Lt, seed=1, T=2)
    parts = [torch.zeros(B, Lt, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(parts, torch.from_numpy(mine["text"]))
    nparts = [torch.zeros(3, B, L, 2) for _ in range(world)]
    dist.all_gather(nparts, torch.from_numpy(mine["noise"]))
    full = spec.synthetic_inputs(world * B, L, Lt, seed=1, T=2)
    ok = np.array_equal(torch.cat(parts).numpy(), full["text"])
    ok &= np.array_equal(torch.cat(nparts, dim=1).numpy(), full["noise"])
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok &= float(t) == float(world)
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_batch_shards_cover_the_global_batch_gloo():
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        assert dict(ret) == {0: True, 1: True}
