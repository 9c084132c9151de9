#!/bin/bash
# One GPU-box pass that produces every profile artefact of a round: rocprofv3 kernel-trace stats, the two HBM PMC passes
# (FETCH_SIZE / WRITE_SIZE, separate runs), the SQ pass, and a plain bench line.  Usage (on the GPU box, from the repo root):
#     bash tools/profile_all.sh r02     -> gpurun_out/prof_r02/...   (copy the summaries into profiles/ afterwards)
set -e
tag=${1:-rXX}
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-fp32 --no-train-step --no-longseq"
cd /tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d "$out/trace" --output-format csv -- $BENCH > "$out/trace.log" 2>&1
echo "[profile_all] kernel trace done"
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE -d "$out/fetch" --output-format csv -- $BENCH > "$out/fetch.log" 2>&1
echo "[profile_all] FETCH_SIZE pass done"
timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE -d "$out/write" --output-format csv -- $BENCH > "$out/write.log" 2>&1
echo "[profile_all] WRITE_SIZE pass done"
timeout -k 10 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES \
  -d "$out/sq" --output-format csv -- $BENCH > "$out/sq.log" 2>&1
echo "[profile_all] SQ pass done"
timeout -k 10 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES \
  -d "$out/insts" --output-format csv -- $BENCH > "$out/insts.log" 2>&1
echo "[profile_all] instruction-mix pass done"
timeout -k 10 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d "$out/l2" --output-format csv -- $BENCH > "$out/l2.log" 2>&1
echo "[profile_all] L2 pass done"
cd - > /dev/null
python3 tools/pmc_traffic.py "$out/fetch" "$out/write" "$out/${tag}_hbm_traffic_pmc.json" > "$out/traffic.txt"
python3 tools/pmc_traffic_by_dispatch.py "$out/fetch" "$out/write" "$out/${tag}_hbm_traffic_by_launch.json" > "$out/${tag}_hbm_traffic_by_launch.txt"
python3 tools/pmc_sq.py "$out/sq" "$out/${tag}_sq_counters.json" > "$out/sq.txt"
python3 tools/pmc_insts.py "$out/insts" "$out/${tag}_inst_mix.json" > "$out/insts.txt"
python3 tools/pmc_l2.py "$out/l2" "$out/${tag}_l2_counters.json" > "$out/l2.txt"
cp $(find "$out/trace" -name "*kernel_stats.csv" | head -1) "$out/${tag}_rocprof_kernel_stats.csv"
python3 bench.py --steps 10 --warmup 3 > "$out/${tag}_bench.json" 2> "$out/bench.err"
# the big per-dispatch CSVs stay on the box; only the summaries are merged back
rm -rf "$out/trace" "$out/fetch" "$out/write" "$out/sq" "$out/insts" "$out/l2"
ls -la "$out"
