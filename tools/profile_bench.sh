#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/profile_bench.sh <tag> [bench args...]
# rocprofv3 kernel trace of bench.py, reduced to the steady-state window -> gpurun_out/<tag>/steady_kernel_stats.csv
set -u
TAG=$1; shift
STEPS=${STEPS:-5}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$TAG
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$TAG -o r -- python bench.py --steps $STEPS --warmup 3 --no-cpu-baseline --no-categories --no-parity --skip-kernel-timing "$@" > gpurun_out/$TAG/bench.log 2>&1
grep '^{"metric"' gpurun_out/$TAG/bench.log > gpurun_out/$TAG/bench.json
MS=$(python -c "import json; d=json.load(open('gpurun_out/$TAG/bench.json')); print(d['ms_per_step']*$STEPS)")
echo "steady window: $MS ms"; cut -c1-400 gpurun_out/$TAG/bench.json
python tools/trace_summary.py /tmp/prof_$TAG/r_kernel_trace.csv --last-ms $MS --steps $STEPS --out gpurun_out/$TAG/steady_kernel_stats.csv --top ${TOP:-70}
