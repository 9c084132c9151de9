#!/bin/bash
# Builds the micro-benchmarks under tools/ into tools/bin/ (git-ignored; the directory travels to the GPU box with gpurun).
# Needs the library objects: python -c "import __graft_entry__ as g; g.build()" first.
set -e
cd "$(dirname "$0")/.."
P="diffusion-handwriting-generation.pytorch_amd/build"
C="diffusion-handwriting-generation.pytorch_amd/csrc"
H="hipcc --offload-arch=gfx950 -O2 -std=c++17"
mkdir -p tools/bin
# the library is built WITHOUT the per-stage stamp code (csrc/dhw_common.h, DHW_STAMPS); the stamp-printing benches get
# their own stamped kernel objects
S=tools/bin/stamped
mkdir -p $S
pids=""
for f in convblock enclayer gemm train; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDHW_STAMPS -x hip -c $C/$f.hip -o $S/$f.o & pids="$pids $!"; done
for pid in $pids; do wait $pid; done   # (a bare `wait` hides a failed compile from `set -e`)
P=$S
$H -c tools/bench_conv.cpp -o tools/bin/bench_conv.o && hipcc --offload-arch=gfx950 tools/bin/bench_conv.o $P/convblock.o $P/enclayer.o -o tools/bin/bench_conv
$H -c tools/bench_enc.cpp -o tools/bin/bench_enc.o && hipcc --offload-arch=gfx950 tools/bin/bench_enc.o $P/enclayer.o -o tools/bin/bench_enc
$H -c tools/bench_sgemm.cpp -o tools/bin/bench_sgemm.o && hipcc --offload-arch=gfx950 tools/bin/bench_sgemm.o $P/train.o -o tools/bin/bench_sgemm
$H -c tools/bench_text.cpp -o tools/bin/bench_text.o && hipcc --offload-arch=gfx950 tools/bin/bench_text.o $P/gemm.o -o tools/bin/bench_text
$H -c tools/bench_encw.cpp -o tools/bin/bench_encw.o && hipcc --offload-arch=gfx950 tools/bin/bench_encw.o $P/enclayer.o -o tools/bin/bench_encw
for t in bench_l2 bench_copy bench_handoff bench_lat; do $H tools/$t.cpp -o tools/bin/$t; done
echo "built: $(ls tools/bin | grep -v '\.o$' | tr '\n' ' ')"
# ablation builds of the ConvBlock bench (diagnostics; DHW_ABL bit mask, see csrc/convblock.hip): tools/bin/bench_conv_abl<N>
if [ -n "$DHW_BUILD_ABL" ]; then
  for abl in $DHW_BUILD_ABL; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDHW_STAMPS -DDHW_ABL=$abl -x hip -c "diffusion-handwriting-generation.pytorch_amd/csrc/convblock.hip" -o /tmp/convblock_abl$abl.o &&
      hipcc --offload-arch=gfx950 tools/bin/bench_conv.o /tmp/convblock_abl$abl.o $P/enclayer.o -o tools/bin/bench_conv_abl$abl
  done
fi
