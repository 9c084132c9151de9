#!/bin/bash
# Build the library as of git HEAD into tools/bin/libdhw_prev.so (baseline for tools/ab.sh), then rebuild the working tree.
set -e
cd "$(dirname "$0")/.."
git stash -q
python "diffusion-handwriting-generation.pytorch_amd/build.py" >/dev/null
cp "diffusion-handwriting-generation.pytorch_amd/libdhw_hip.so" tools/bin/libdhw_prev.so
git stash pop -q
python "diffusion-handwriting-generation.pytorch_amd/build.py" >/dev/null
ls -la tools/bin/libdhw_prev.so "diffusion-handwriting-generation.pytorch_amd/libdhw_hip.so"
