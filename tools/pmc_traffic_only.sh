#!/bin/bash
# The two HBM PMC passes of tools/profile_all.sh alone (FETCH_SIZE / WRITE_SIZE, separate runs) + their summaries:
#     bash tools/pmc_traffic_only.sh r05   -> gpurun_out/prof_r05/r05_hbm_traffic_pmc.json, r05_hbm_traffic_by_launch.{json,txt}
set -e
tag=${1:-rXX}
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-fp32 --no-train-step --no-longseq"
cd /tmp
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE -d "$out/fetch" --output-format csv -- $BENCH > "$out/fetch.log" 2>&1
timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE -d "$out/write" --output-format csv -- $BENCH > "$out/write.log" 2>&1
cd - > /dev/null
python3 tools/pmc_traffic.py "$out/fetch" "$out/write" "$out/${tag}_hbm_traffic_pmc.json" > "$out/traffic.txt"
python3 tools/pmc_traffic_by_dispatch.py "$out/fetch" "$out/write" "$out/${tag}_hbm_traffic_by_launch.json" > "$out/${tag}_hbm_traffic_by_launch.txt"
rm -rf "$out/fetch" "$out/write"
cat "$out/${tag}_hbm_traffic_by_launch.txt"
