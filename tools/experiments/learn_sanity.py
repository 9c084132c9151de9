import os, sys, math, tempfile, torch, yaml, numpy as np
sys.path.insert(0, os.getcwd())
import dhg_amd
from dhg_amd import train_model as tm
# 16 smooth synthetic strokes (L=64): sums of sines; batch 16, warm-up 1000
g = torch.Generator().manual_seed(0)
N, L, Lt = 16, 64, 10
t = torch.linspace(0, 1, L)[None, :, None]
f = torch.rand(N, 1, 2, generator=g) * 4 + 1
ph = torch.rand(N, 1, 2, generator=g) * 6.28
strokes = torch.sin(6.28 * f * t + ph) * 0.5
pen = (torch.rand(N, L, 1, generator=g) < 0.05).float()
data = {"strokes": torch.cat([strokes, pen], -1), "text": torch.randint(1, 73, (N, Lt), generator=g), "style": torch.randn(N, 14, 1280, generator=g)}
d = tempfile.mkdtemp()
torch.save(data, os.path.join(d, "batches.pt"))
cfg = {"training_args": {"steps": 600, "batch_size": 16, "warmup_steps": 1000, "clip_grad": 100.0, "dropout": 0.0, "att_layers_num": 2, "channels": 128, "log_freq": 150, "save_freq": 100000},
       "dataset_args": {"max_seq_len": L, "max_text_len": Lt}, "optimizer": {"params": {"betas": [0.9, 0.98], "weight_decay": 1e-5}}}
yaml.safe_dump(cfg, open(os.path.join(d, "cfg.yml"), "w"))
tm.fit(os.path.join(d, "cfg.yml"), os.path.join(d, "batches.pt"), os.path.join(d, "out"), seed=0)
