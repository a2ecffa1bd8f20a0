#!/bin/bash
# same-box A/B of the Swin configs: tools/ab_swin.sh <swinb|swinl|swinl_mask2former_fp8> <size> <steps> "<label>|ENV=.. ENV=.." ...
NAME=$1; SIZE=$2; STEPS=$3; shift; shift; shift
for spec in "$@"; do
  label="${spec%%|*}"; rest="${spec#*|}"
  env PD_CONFIG=$NAME $rest python tools/bench_config3.py $SIZE $STEPS 2>/dev/null | grep '^{"workload' > /tmp/ab_swin.json
  python - "$label" <<'PY'
import json, sys
d = json.load(open('/tmp/ab_swin.json'))
print(f"{sys.argv[1]:28s} {d['ms_per_step']:.2f} ms  host issue {d['host_issue_ms']:.2f}  loss {d['loss']:.4f}")
PY
done
