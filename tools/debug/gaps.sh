#!/bin/bash
# Usage (GPU box): tools/debug/gaps.sh <tag>: idle gaps between kernels of the last step under rocprofv3 --kernel-trace -> gpurun_out/<tag>/timeline.txt
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$TAG; rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$TAG -o r -- python bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-categories --no-parity --skip-kernel-timing "$@" > gpurun_out/$TAG/bench.log 2>&1
MS=$(python -c "import json; d=json.loads([l for l in open('gpurun_out/$TAG/bench.log') if l.startswith('{\"metric')][0]); print(d['ms_per_step'])")
python tools/trace_timeline.py /tmp/prof_$TAG/r_kernel_trace.csv --last-ms $MS --out gpurun_out/$TAG/timeline.txt | tail -1
echo "step under the profiler: $MS ms"
