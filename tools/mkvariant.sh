#!/bin/bash
# tools/mkvariant.sh <name> <file.hip> "<-D flags>" — a variant of the library with ONE translation unit rebuilt with extra flags: tools/bin/libdhw_<name>.so
# (the other objects come from the package's build directory: run the normal build first).  For same-box A/B runs (tools/ab2.sh, DHW_LIB).
set -e
cd "$(dirname "$0")/.."
P="diffusion-handwriting-generation.pytorch_amd"
name=$1; src=$2; flags=$3
base=$(basename "$src" | sed 's/\.[a-z]*$//')
mkdir -p tools/bin/variants
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value $flags -x hip -c "$P/csrc/$src" -o "tools/bin/variants/${base}_$name.o"
objs=""
for o in gemm convblock enclayer persist attn misc style textside train dhw_api dhw_style_api dhw_train_api; do
  if [ "$o" = "$base" ]; then objs="$objs tools/bin/variants/${base}_$name.o"; else objs="$objs $P/build/$o.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o "tools/bin/libdhw_$name.so" $objs
echo "tools/bin/libdhw_$name.so"
