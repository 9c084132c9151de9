#!/bin/bash
# two resident workgroups per CU in different stages, built once on one kernel and measured.
# enc1 / enc2 ConvBlocks, 62-row tiles: one 8-wave workgroup per CU (256 VGPRs) vs TWO co-resident 8-wave workgroups per CU
# (128 VGPRs, 80 KB LDS each: DHW_CONV_OCC=2), the second half of the grid started 0 / 1 / 2 us late (DHW_CONV_STAGGER, in 0.5 us
# units) so that co-resident workgroups sit in different stages.  Per variant: time per launch (events) and the SQ counters that
# say whether matrix and vector work of the two workgroups ran together (SQ_VALU_MFMA_COEXEC_CYCLES, SQ_VALU_MFMA_BUSY_CYCLES).
# Run on the GPU box from the repo root: bash tools/two_phase_experiment.sh > gpurun_out/<dir>/two_phase.log
export TMPDIR=/tmp
R=$PWD
run() {   # label, env...
  local label=$1; shift
  echo "== $label"
  env "$@" BENCH_ONLY=enc $R/tools/bin/bench_conv 50 2>&1 | grep "^enc1\|^enc2" | cut -c1-60
  (cd /tmp && env "$@" timeout -k 10 240 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY \
      -d /tmp/tp_$$ --output-format csv -- $R/tools/bin/bench_conv 10 > /dev/null 2>&1)
  python3 - /tmp/tp_$$ <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if "convblock" not in k: continue
        k = k.split("convblock_kernel")[1][:44]
        if not (k.startswith("IDF16bLi64ELi128ELi8ELi1ELi0ELi0ELi128E") or k.startswith("IDF16bLi64ELi192ELi8ELi1ELi0ELi0ELi128E") or "ELi8ELi2E" in k): continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_BUSY_CYCLES": n[k] += 1
for k in sorted(tot):
    c = tot[k]; m = n[k]
    print(f"   {k}: launches {m}, per launch: busy {c['SQ_BUSY_CYCLES']/m:.0f}, MFMA busy {c['SQ_VALU_MFMA_BUSY_CYCLES']/m:.0f}, MFMA+VALU co-exec {c['SQ_VALU_MFMA_COEXEC_CYCLES']/m:.0f} "
          f"({100*c['SQ_VALU_MFMA_COEXEC_CYCLES']/max(c['SQ_VALU_MFMA_BUSY_CYCLES'],1):.1f} % of MFMA busy), VALU active {c['SQ_ACTIVE_INST_VALU']/m:.0f}, VALU insts {c['SQ_INSTS_VALU']/m:.0f}, MFMA insts {c['SQ_INSTS_MFMA']/m:.0f}")
PY
  rm -rf /tmp/tp_$$
}
run "one 8-wave workgroup per CU, 62-row tiles (DHW_CONV_BM=64)" DHW_CONV_BM=64
run "two co-resident workgroups per CU, no stagger" DHW_CONV_BM=64 DHW_CONV_OCC=2 DHW_CONV_STAGGER=0
run "two co-resident workgroups per CU, second half 1 us late" DHW_CONV_BM=64 DHW_CONV_OCC=2 DHW_CONV_STAGGER=2
run "two co-resident workgroups per CU, second half 2 us late" DHW_CONV_BM=64 DHW_CONV_OCC=2 DHW_CONV_STAGGER=4
run "two co-resident workgroups per CU, second half 4 us late" DHW_CONV_BM=64 DHW_CONV_OCC=2 DHW_CONV_STAGGER=8
