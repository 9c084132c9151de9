#!/bin/bash
# A/B two builds of libdhw_hip.so on ONE GPU box: tools/bin/libdhw_prev.so (baseline) vs the in-tree build, alternating.
for i in 1 2 3; do
  for lib in tools/bin/libdhw_prev.so ""; do
    r=$(DHW_LIB=${lib:+$PWD/$lib} timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-profile --steps 10 2>&1 | grep -o "[0-9.]* ms/step")
    echo "${lib:-new}: $r"
  done
done
