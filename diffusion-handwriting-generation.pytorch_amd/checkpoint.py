"""Checkpoint / config front end of the sampler: what the reference does between the command line and
the sampling loop (reference checkpoint.py:92-130 `load_checkpoint`, :256-297 `load_model`; inference.py:28-58 checkpoint
discovery).  Files are read with loaders that execute nothing from the file (`torch.load(weights_only=True)`,
`yaml.safe_load`)."""
from __future__ import annotations

import re
from pathlib import Path

import torch

from .model import DiffusionModel


def read_state_dict(filename: str | Path) -> dict:
    """The 323-key state_dict of a reference checkpoint: a bare dict or ``{"state_dict": ...}``, keys optionally
    prefixed with ``module.`` (DataParallel), reference checkpoint.py:117-129."""
    ckpt = torch.load(str(filename), map_location="cpu", weights_only=True)
    if not isinstance(ckpt, dict):
        raise RuntimeError(f"No state_dict found in checkpoint file {filename}")
    sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
    return {re.sub(r"^module\.", "", k): v for k, v in sd.items()}


def find_checkpoint(experiment_path: str | Path) -> Path | None:
    """model_final.pth, else model_last.pth, else the checkpoint_<step>.pth with the largest integer step
    (reference inference.py:32-52)."""
    exp = Path(experiment_path)
    for name in ("model_final.pth", "model_last.pth"):
        if (exp / name).exists():
            return exp / name
    numbered = []
    for p in exp.glob("checkpoint_*.pth"):
        try:
            numbered.append((int(p.stem.split("_")[1]), p))
        except ValueError:
            continue
    return max(numbered)[1] if numbered else None


def read_config(config_path: str | Path) -> dict:
    """Model hyper-parameters of a reference experiment config (configs/best.yml:25-27): ``training_args.att_layers_num``,
    ``.channels``, ``.dropout`` -> DiffusionModel(num_layers, c1, c2 = 3*c1/2, c3 = 2*c1, drop_rate), checkpoint.py:280-286."""
    import yaml

    with open(config_path) as f:
        cfg = yaml.safe_load(f) or {}
    ta = cfg.get("training_args")
    if not isinstance(ta, dict):
        raise KeyError(f"{config_path}: no `training_args` section")
    missing = [k for k in ("att_layers_num", "channels", "dropout") if k not in ta]
    if missing:   # the reference reads cfg.training_args.<key> and fails on an incomplete config; it has no defaults
        raise KeyError(f"{config_path}: training_args is missing {', '.join(missing)}")
    ch = int(ta["channels"])
    return {"num_layers": int(ta["att_layers_num"]), "c1": ch, "c2": ch * 3 // 2, "c3": ch * 2,
            "drop_rate": float(ta["dropout"])}


def load_model(config_path: str | Path | None = None, checkpoint_path: str | Path | None = None, *, precision: str = "bf16",
               **capacity) -> DiffusionModel:
    """``load_model`` of the reference (checkpoint.py:256-297): build the model the config describes (or, without a
    config, the one the checkpoint's attention-layer count implies) and strict-load the checkpoint."""
    sd = read_state_dict(checkpoint_path) if checkpoint_path is not None else None
    if config_path is not None:
        dims = read_config(config_path)
    else:
        if sd is None:
            raise ValueError("load_model needs a config_path or a checkpoint_path")
        layers = {int(m.group(1)) for k in sd for m in [re.match(r"att_layers\.(\d+)\.", k)] if m}
        dims = {"num_layers": max(layers) + 1 if layers else 0, "c1": 128, "c2": 192, "c3": 256, "drop_rate": 0.0}
    model = DiffusionModel(dims["num_layers"], dims["c1"], dims["c2"], dims["c3"], dims["drop_rate"], precision=precision,
                           **capacity).eval()
    if sd is not None:
        model.load_state_dict(sd, strict=True)   # RuntimeError naming missing / unexpected keys, as the reference
    return model
