#!/bin/bash
# Evidence for / against the ConvBlock ping-pong variant (csrc/convblock_core.h, PP; DHW_CONV_PP): per-wave stage stamps of one
# workgroup (stamped build, tools/bench_conv) and the SQ co-execution counters, for the lockstep and the ping-pong schedule.
# usage (GPU box, repo root): bash tools/experiments/conv_pp_evidence.sh <tag>   -> gpurun_out/<tag>/
set -e
tag=${1:-pp}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
bash tools/build_tools.sh > "$out/build_tools.log" 2>&1
for pp in 0 3; do
  STAMP_WAVES=1 DHW_CONV_PP=$pp timeout -k 10 120 tools/bin/bench_conv 50 > "$out/bench_conv_pp$pp.log" 2>&1
done
export TMPDIR=/tmp
for pp in 0 3; do
  (cd /tmp && DHW_CONV_PP=$pp timeout -k 10 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY \
     -d "$out/pmc_pp$pp" --output-format csv -- $OLDPWD/tools/bin/bench_conv 5 > "$out/pmc_pp$pp.log" 2>&1)
  python3 - "$out/pmc_pp$pp" > "$out/coexec_pp$pp.txt" <<'PY'
import csv, glob, os, sys, re
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int))
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        k = row["Kernel_Name"]
        if "convblock_kernel" not in k: continue
        # template arguments <T, BM, CO, NW, OCC, UPC, CH, CIN, TIGHT, PP> from the demangled or the mangled (…ILi128ELi128E…) name
        m = re.search(r"convblock_kernel<([^>]*)>", k)
        key = m.group(1) if m else ",".join(re.findall(r"Li(\d+)E", k))
        acc[key][row["Counter_Name"]] += float(row["Counter_Value"]); n[key][row["Counter_Name"]] += 1
for k in sorted(acc):
    a = {c: acc[k][c] / max(n[k][c], 1) for c in acc[k]}
    busy, co = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), a.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0)
    print(f"convblock_kernel<{k}>: launches {max(n[k].values())}  mfma_busy {busy:.3e}  coexec {co:.3e}  coexec/mfma_busy {co / busy if busy else 0:.3f}  "
          f"wave_cycles {a.get('SQ_WAVE_CYCLES', 0):.3e}  wait_any {a.get('SQ_WAIT_ANY', 0):.3e}  insts valu {a.get('SQ_INSTS_VALU', 0):.3e} mfma {a.get('SQ_INSTS_MFMA', 0):.3e}")
PY
  rm -rf "$out/pmc_pp$pp"
done
tail -n +1 "$out"/coexec_pp*.txt
grep -h "us/launch" "$out"/bench_conv_pp*.log
