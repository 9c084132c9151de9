#!/bin/bash
# Build the library as of git HEAD into tools/bin/libdhw_prev.so (baseline for tools/ab.sh), then rebuild the working tree.
# The working tree is stashed only when it differs from HEAD (a bare `git stash; git stash pop` on a clean tree would pop an
# unrelated older stash entry).
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin
stashed=0
if ! git diff --quiet || ! git diff --cached --quiet; then
  git stash push -q -m mkprev-$$
  stashed=1
fi
restore() { if [ "$stashed" = 1 ]; then git stash pop -q; stashed=0; fi; }
trap restore EXIT
python "diffusion-handwriting-generation.pytorch_amd/build.py" >/dev/null
cp "diffusion-handwriting-generation.pytorch_amd/libdhw_hip.so" tools/bin/libdhw_prev.so
restore
python "diffusion-handwriting-generation.pytorch_amd/build.py" >/dev/null
ls -la tools/bin/libdhw_prev.so "diffusion-handwriting-generation.pytorch_amd/libdhw_hip.so"
