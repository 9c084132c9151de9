#!/bin/bash
# tools/gpu_iter.sh <tag> '<command run on the GPU box>' — rebuild the library + micro-benchmarks, run the command on an MI355X box
# (output under gpurun_out/<tag>/), print the tail.  One edit-measure iteration of the kernel work.
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
python -c "import __graft_entry__ as g; g.build()"
bash tools/build_tools.sh > /tmp/build_tools.log 2>&1 || { tail -20 /tmp/build_tools.log; exit 1; }
/usr/local/graft/bin/gpurun --timeout ${GPU_TIMEOUT:-900} -- "mkdir -p gpurun_out/$tag; $*" 2>&1 | grep -v "^\[gpurun\] sending" | tail -${TAIL:-60}
