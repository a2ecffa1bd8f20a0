#!/bin/bash
# same as tools/ab_bench.sh with 80 timed steps after 10 (for differences below 0.1 ms)
for spec in "$@"; do
  label="${spec%%|*}"; rest="${spec#*|}"
  env $(echo "$rest" | tr ' ' '\n' | grep '=' | grep -v '^--' | tr '\n' ' ') python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-categories --no-parity --skip-kernel-timing $(echo "$rest" | tr ' ' '\n' | grep -v '=' | tr '\n' ' ') 2>/dev/null | grep '^{"metric' > /tmp/ab.json
  python - "$label" <<'PY'
import json, sys
d = json.load(open('/tmp/ab.json'))
s = d["ms_per_step_stats"]
print(f"{sys.argv[1]:28s} mean {d['ms_per_step']:.3f} ms  median {s['median']:.3f}  p10 {s['p10']:.3f}  p90 {s['p90']:.3f}")
PY
done
