"""Helper of test_gpu_train_model.py::test_two_rank_update_equals_the_single_rank_update_on_the_whole_batch: one rank of a
data-parallel training run on a SHARED GPU (gloo rendezvous on 127.0.0.1; the gradient all-reduce is the product's
bucketed ``GradBucketReducer`` behind the graph segments of ``GraphedTrainStep``; DHW_TRAIN_BUCKETS=0: the flat ``allreduce_grads``).  Each rank takes its slice of one seeded global batch; rank 0 writes the parameters after two updates."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    torch.set_num_threads(4)
    from dhg_amd import spec, train, train_model as tm
    if world > 1:
        torch.distributed.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    Bg, L, Lt, S = 4, 64, 10, 14
    B = Bg // world
    sl = slice(rank * B, (rank + 1) * B)
    sd = spec.synthetic_state_dict(2, 128, 192, 256, seed=0)
    model = tm.TrainModel(sd, num_layers=2, device=dev)
    opt = train.Adam(model.parameters())
    inp = spec.synthetic_inputs(Bg, L, Lt, S=S, seed=51, pad=2)
    g = torch.Generator().manual_seed(51)
    pen = (torch.rand(Bg, L, 1, generator=g) < 0.1).float()
    eps = torch.randn(Bg, L, 2, generator=g)
    alphas = torch.rand(Bg, 1, generator=g) * 0.9 + 0.05
    keep = (torch.rand(2, Bg, S, 1280, generator=g) >= 0.3).float()
    batch = {"strokes": torch.cat([torch.from_numpy(inp["strokes"]), pen], dim=-1)[sl], "text": torch.from_numpy(inp["text"])[sl],
             "style": torch.from_numpy(inp["style"])[sl]}
    step = tm.GraphedTrainStep(model, opt, B, L, Lt, S, warmup=100, device_rng=False)
    losses = []
    for k in (1, 2):
        losses.append(step(batch, None, k, eps=eps[sl], alphas=alphas[sl], style_keep=keep[k - 1, sl]).cpu().numpy())
    torch.cuda.synchronize()
    if rank == 0:
        # (the parameters in state_dict order and torch layouts: the flat buffer's own order is the gradient buckets', private to the trainer)
        flat = torch.cat([v.detach().contiguous().reshape(-1) for v in model.state_dict().values()])
        np.savez(out, flat=flat.cpu().numpy(), losses=np.array(losses), segments=0 if step.segments is None else len(step.segments))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
