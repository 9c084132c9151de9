#!/bin/bash
# tools/ab2.sh <rounds> <libA> <libB> ... — alternate bench runs of several builds of libdhw_hip.so on ONE GPU box ("" or "new" = the in-tree build).
# Every GPU step runs under `timeout -k 10` (round 4: the one rocprofv3 call here without it sat stuck for 14 minutes after an abort).
# Prints ms per 60-step batch per run, then (STATS=1) one rocprofv3 kernel-trace summary per build under gpurun_out/$TAG/.
rounds=$1; shift
TAG=${TAG:-ab}
mkdir -p gpurun_out/$TAG
for i in $(seq 1 $rounds); do
  for lib in "$@"; do
    L=""; [ "$lib" != "new" ] && [ -n "$lib" ] && L=$PWD/$lib
    r=$(DHW_LIB=$L timeout -k 10 300 python bench.py --no-cpu-baseline --no-kernel-profile --no-fp32 --no-train-step --no-longseq --steps ${STEPS:-20} 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
    echo "round $i ${lib:-new}: $r" | tee -a gpurun_out/$TAG/ab.log
  done
done
if [ -n "$STATS" ]; then
  export TMPDIR=/tmp
  for lib in "$@"; do
    L=""; [ "$lib" != "new" ] && [ -n "$lib" ] && L=$PWD/$lib
    name=$(basename "${lib:-new}" .so)
    out=$PWD/gpurun_out/$TAG/trace_$name
    (cd /tmp && DHW_LIB=$L timeout -k 10 240 rocprofv3 --kernel-trace --stats -d "$out" --output-format csv -- python3 $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-fp32 --no-train-step --no-longseq > "$out.log" 2>&1)
    cp $(find "$out" -name "*kernel_stats.csv" | head -1) gpurun_out/$TAG/stats_$name.csv
    rm -rf "$out"
    echo "== $name"; head -16 gpurun_out/$TAG/stats_$name.csv | cut -d, -f1-4 | sed 's/_ZN12_GLOBAL__N_1//'
  done
fi
