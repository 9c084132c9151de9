#!/bin/bash
# tools/mkvariant_all.sh <name> "<-D flags>" — a variant of the library with EVERY sampler kernel translation unit rebuilt with extra
# flags (switches that live in shared headers, e.g. -DDHW_APF=0): tools/bin/libdhw_<name>.so.  The host objects come from the package's
# build directory (run the normal build first).  For same-box A/B runs (tools/ab2.sh, DHW_LIB).
set -e
cd "$(dirname "$0")/.."
P="diffusion-handwriting-generation.pytorch_amd"
name=$1; flags=$2
mkdir -p tools/bin/variants
pids=""
for base in gemm convblock enclayer persist attn textside; do
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value $flags -x hip -c "$P/csrc/$base.hip" -o "tools/bin/variants/${base}_$name.o" & pids="$pids $!"
done
for pid in $pids; do wait $pid; done
objs=""
for o in gemm convblock enclayer persist attn misc style textside train dhw_api dhw_style_api dhw_train_api; do
  if [ -f "tools/bin/variants/${o}_$name.o" ]; then objs="$objs tools/bin/variants/${o}_$name.o"; else objs="$objs $P/build/$o.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o "tools/bin/libdhw_$name.so" $objs
echo "tools/bin/libdhw_$name.so"
