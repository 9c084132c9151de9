#!/bin/bash
# tools/regs.sh <file.hip> [grep-pattern] — VGPR / spill / scratch per kernel of one translation unit (compile only)
cd "$(dirname "$0")/../diffusion-handwriting-generation.pytorch_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 ${DHW_DEFS} -x hip -c "$1" -o /tmp/regs_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk '/Function Name:/ {n=$(NF-1)} / VGPRs:/ {v=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /VGPRs Spill/ {print "vgpr", v, "scratch", s, "spill", $(NF-1), n}' |
  grep -E "${2:-.}"
rm -f /tmp/regs_$$.o
