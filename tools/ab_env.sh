#!/bin/bash
# tools/ab_env.sh <rounds> "<ENV=a>" "<ENV=b>" ... — alternate bench runs of the in-tree library under different environments (switches read at dhw_create)
rounds=$1; shift
TAG=${TAG:-abenv}
mkdir -p gpurun_out/$TAG
for i in $(seq 1 $rounds); do
  for e in "$@"; do
    r=$(env $e timeout -k 10 300 python bench.py --no-cpu-baseline --no-kernel-profile --no-fp32 --no-train-step --no-longseq --steps ${STEPS:-20} 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
    echo "round $i [$e]: $r" | tee -a gpurun_out/$TAG/ab.log
  done
done
