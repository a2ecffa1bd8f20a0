#!/bin/bash
# Usage (GPU box, repo root): tools/profile_config3.sh <tag> <swinb|swinl> <size>
# rocprofv3 kernel trace of tools/bench_config3.py (BASELINE configs 3 / 5 on one GPU), reduced to the steady-state window
# -> gpurun_out/<tag>/steady_kernel_stats.csv + bench.json (+ list.txt: every launch of the last step matching $LIST)
set -u
TAG=$1; NAME=$2; SIZE=$3
STEPS=${STEPS:-6}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$TAG
rm -rf /tmp/prof_$TAG
PD_CONFIG=$NAME rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$TAG -o r -- python tools/bench_config3.py $SIZE $STEPS > gpurun_out/$TAG/bench.log 2>&1
grep '^{"workload"' gpurun_out/$TAG/bench.log > gpurun_out/$TAG/bench.json
MS=$(python -c "import json; d=json.load(open('gpurun_out/$TAG/bench.json')); print(d['ms_per_step']*$STEPS)")
echo "steady window: $MS ms"; cat gpurun_out/$TAG/bench.json
python tools/trace_summary.py /tmp/prof_$TAG/r_kernel_trace.csv --last-ms $MS --steps $STEPS --out gpurun_out/$TAG/steady_kernel_stats.csv --top ${TOP:-40}
[ -n "${LIST:-}" ] && python tools/trace_summary.py /tmp/prof_$TAG/r_kernel_trace.csv --last-ms $MS --steps $STEPS --top 0 --list "$LIST" > gpurun_out/$TAG/list.txt
