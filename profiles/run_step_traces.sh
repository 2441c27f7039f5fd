#!/bin/bash
# bash profiles/run_step_traces.sh r02 : one traced training step per workload -> profiles/<tag>/step_<w>.txt
set -o pipefail
TAG=${1:-r02}
export TMPDIR=/tmp
mkdir -p profiles/$TAG
for w in ${2:-c2 c2adam c4s c4 c4g}; do
  OUT=gpurun_out/trace_${TAG}_$w
  rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 profiles/step_trace.py run $w > $OUT.log 2>&1 || { tail -20 $OUT.log; exit 1; }
  python3 profiles/step_trace.py show $OUT > gpurun_out/step_${TAG}_$w.txt || exit 1
done
