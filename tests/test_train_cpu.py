"""Host logic of the training step's first slice, pinned by fixtures the imported reference generated
(oracle/make_golden_r2.py): get_alphas' RNG call order, the Noam schedule, and the world-size-2 gradient all-reduce (gloo)."""
import os
import socket

import numpy as np
import pytest
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


BEST_YML = """
experiment: {name: x, seed: 54321}
dataset_args: {max_seq_len: 480, max_text_len: 50, img_height: 96, img_width: 1400}
training_args:
  steps: 60000
  batch_size: 96
  warmup_steps: 10000
  clip_grad: 100.0
  dropout: 0.0
  att_layers_num: 2
  channels: 128
  log_freq: 5
  save_freq: 1000
optimizer:
  type: torch.optim.Adam
  params: {lr: 0.0003, weight_decay: 0.00001, betas: [0.9, 0.98]}
"""


def test_train_config_keys_of_the_reference_yml(tmp_path):
    """configs/best.yml's training keys (train.py:136-151) -> what fit() uses; an incomplete config raises like the reference's
    attribute access does."""
    from dhg_amd import train_model
    p = tmp_path / "best.yml"
    p.write_text(BEST_YML)
    c = train_model.read_train_config(p)
    assert (c["steps"], c["batch_size"], c["warmup"], c["clip_grad"], c["dropout"]) == (60000, 96, 10000, 100.0, 0.0)
    assert (c["num_layers"], c["c1"], c["c2"], c["c3"], c["L"], c["Lt"]) == (2, 128, 192, 256, 480, 50)
    assert c["betas"] == (0.9, 0.98) and c["weight_decay"] == 1e-5 and c["seed"] == 54321
    bad = tmp_path / "bad.yml"
    bad.write_text(BEST_YML.replace("  warmup_steps: 10000\n", ""))
    with pytest.raises(KeyError, match="warmup_steps"):
        train_model.read_train_config(bad)


def _bucket_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(7)
    n = 10007
    full = torch.randn(world, n, generator=g)                   # per-rank flat gradient buffers, known to all
    ranges = [(0, 3000), (3000, 3004), (3004, 9000), (9000, 10000), (10000, n)]
    mine = full[rank].clone()
    red = train.GradBucketReducer(mine, ranges)
    ok = True
    for i in range(len(ranges) - 1):                            # in completion order, as the tape's markers fire
        red.launch(i)
    try:                                                        # waiting with a bucket missing is an error, not a silent partial reduce
        red.wait()
        ok = False
    except RuntimeError:
        pass
    red.launch(len(ranges) - 1)
    try:
        red.launch(2)
        ok = False
    except RuntimeError:
        pass
    red.wait(average=True)
    flat = full[rank].clone()
    train.allreduce_grads([flat])                               # the single flat all-reduce of rounds 2-4
    ok = ok and torch.equal(mine, flat) and torch.allclose(mine, full.mean(0), atol=1e-6)
    red.launch(0)                                               # the reducer is reusable update after update
    for i in range(1, len(ranges)):
        red.launch(i)
    red.wait()
    ok = ok and torch.allclose(mine, full.mean(0) * world, atol=1e-5)
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_bucketed_gradient_allreduce_equals_the_flat_one_gloo():
    """train.GradBucketReducer (the all-reduce overlapped with the backward's tail, bucket by bucket) over a world-size-2 gloo
    group: the buckets tile the flat buffer, each is reduced exactly once per update, and the result is the flat all-reduce's,
    bit for bit."""
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_bucket_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        assert dict(ret) == {0: True, 1: True}
    with pytest.raises(ValueError):
        train.GradBucketReducer(torch.zeros(10), [(0, 4), (5, 10)])
