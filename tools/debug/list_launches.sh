#!/bin/bash
# Usage (GPU box): tools/debug/list_launches.sh <tag> <regex> [bench args]: every launch of the last step whose kernel matches, in order
TAG=$1; PAT=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$TAG; rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$TAG -o r -- python bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-categories --no-parity --skip-kernel-timing "$@" > gpurun_out/$TAG/bench.log 2>&1
MS=$(python -c "import json; d=json.loads([l for l in open('gpurun_out/$TAG/bench.log') if l.startswith('{\"metric')][0]); print(d['ms_per_step']*4)")
python tools/trace_summary.py /tmp/prof_$TAG/r_kernel_trace.csv --last-ms $MS --steps 4 --top 0 --list "$PAT" > gpurun_out/$TAG/list.txt
