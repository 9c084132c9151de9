#!/usr/bin/env python3
"""Command-line sampler with the reference's `infer` arguments (reference inference.py:19-27, `make infer`):

    python infer.py "Follow the White Rabbit" style.npy --experiment-path data/best_exp --output result

`source` is a handwriting image of the writer (as in the reference: cropped, resized to 96 rows, MobileNetV2 StyleExtractor;
`--style-weights` = a local copy of torchvision's mobilenet_v2 checkpoint) or a file with the writer-style features
([14,1280], .npy or .pt)."""
import argparse

import dhg_amd


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("prompt")
    ap.add_argument("source")
    ap.add_argument("--config-path")
    ap.add_argument("--checkpoint-path")
    ap.add_argument("--experiment-path")
    ap.add_argument("--output", default="result")
    ap.add_argument("--diffusion-mode", default="new", choices=["new", "standard"])
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--style-weights", help="torchvision mobilenet_v2 state_dict (.pth) for the StyleExtractor")
    a = ap.parse_args(argv)
    strokes = dhg_amd.infer_file(a.prompt, a.source, a.config_path, a.checkpoint_path, a.experiment_path, a.output,
                                 a.diffusion_mode, precision=a.precision, seed=a.seed, style_weights=a.style_weights)
    print(f"{strokes.shape[0]} stroke points -> ./{a.output}.png")


if __name__ == "__main__":
    main()
