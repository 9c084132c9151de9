#!/usr/bin/env python3
"""Command-line trainer with the reference's `train` arguments (reference train.py:183-189, `make train`):

    python train.py --config diffusion_handwriting_generation/configs/best.yml [--data batches.pt] [--out runs/exp]

Runs the reference's TrainingLoop (train.py:84-134) with every update on the MI355X (dhg_amd.train_model: one replayed
hipGraph per update; under torch.distributed.run one process per GPU with an RCCL all-reduce of the gradients).  The config
is the reference's yml (training_args: steps, batch_size, warmup_steps, clip_grad, dropout, att_layers_num, channels,
log_freq, save_freq; dataset_args: max_seq_len, max_text_len; optimizer.params: betas, weight_decay).  `--data`: a torch file
{"strokes" [N,L,3], "text" [N,Lt] int64, "style" [N,14,1280]} of preprocessed samples (the reference's IAMDataset items,
dataset.py:143-157; read with weights_only=True); without it, synthetic batches of the configured shape (smoke / benchmark
runs — the IAM corpus and its preprocessing are outside this package).  Writes checkpoint_<n>.pth in save_checkpoint's form
({"meta": None, "state_dict": ...}, reference checkpoint.py:244) and model_final.pth as the bare state_dict (train.py:131),
both with the reference's 323 keys; cadence and numbering follow the reference ((count + 1) % freq, see fit()).
--precision bf16 selects the mixed-precision GEMMs (fp32 master weights and optimizer state); batch_size is per rank."""
import argparse

import dhg_amd


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--config", required=True)
    ap.add_argument("--data")
    ap.add_argument("--out", default="runs/exp")
    ap.add_argument("--steps", type=int, help="override training_args.steps")
    ap.add_argument("--init", help="state_dict to start from (.pth)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"], help="GEMM operand precision (fp32 = the reference's)")
    a = ap.parse_args(argv)
    dhg_amd.train_model.fit(a.config, a.data, a.out, steps=a.steps, init=a.init, seed=a.seed, precision=a.precision)


if __name__ == "__main__":
    main()
