#!/bin/bash
# same-box A/B of bench.py variants: tools/ab_bench.sh "<label>|<extra args / env>" ...   (each run: 30 steps after 8 warm-up)
for spec in "$@"; do
  label="${spec%%|*}"; rest="${spec#*|}"
  env $(echo "$rest" | tr ' ' '\n' | grep '=' | grep -v '^--' | tr '\n' ' ') python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-categories --no-parity --skip-kernel-timing $(echo "$rest" | tr ' ' '\n' | grep -v '=' | tr '\n' ' ') 2>/dev/null | grep '^{"metric' > /tmp/ab.json
  python - "$label" <<'PY'
import json, sys
d = json.load(open('/tmp/ab.json'))
s = d["ms_per_step_stats"]
print(f"{sys.argv[1]:28s} mean {d['ms_per_step']:.2f} ms  median {s['median']:.2f}  min {s['min']:.2f}  max {s['max']:.2f}  host issue {d['config']['host_issue_ms_per_step']:.2f}  host cpu {d['config']['host_cpu_ms_per_step']:.1f}")
PY
done
