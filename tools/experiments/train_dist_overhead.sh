#!/bin/bash
# usage (GPU box): bash tools/experiments/train_dist_overhead.sh > gpurun_out/<tag>/train_dist_overhead.log
for r in 1 2; do
  for m in plain eager_adam seg_noreduce seg_bucketed flat; do
    timeout -k 10 120 python tools/experiments/train_dist_overhead.py $m 2>/dev/null | grep "ms per update"
  done
done
