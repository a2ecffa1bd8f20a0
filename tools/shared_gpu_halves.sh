#!/bin/bash
# Potential of feeding ONE GPU from two host processes that each train ONE image of the batch of two (no gradient exchange: the upper
# bound of what a two-process form of the step could reach).  Usage (GPU box, repo root): tools/shared_gpu_halves.sh -> gpurun_out/shared_halves.txt
OUT=gpurun_out/shared_halves.txt; mkdir -p gpurun_out; : > $OUT
run() {   # n processes, batch b
  pids=""
  for r in $(seq 1 $1); do
    python bench.py --gpus 1 --steps 40 --warmup 10 --batch $2 --no-cpu-baseline --no-categories --no-parity --skip-kernel-timing 2>/dev/null | grep '^{"metric' > /tmp/sh_$r.json &
    pids="$pids $!"
  done
  for p in $pids; do wait $p; done
  python - $1 $2 >> $OUT <<'PY'
import json, sys
n, b = int(sys.argv[1]), int(sys.argv[2])
ds = [json.load(open(f'/tmp/sh_{r}.json')) for r in range(1, n + 1)]
ms = [d['ms_per_step'] for d in ds]
print(f"{n} process(es) x batch {b}: {min(ms):.2f} .. {max(ms):.2f} ms per step each, {sum(d['value'] for d in ds):.1f} images/s together; host issue {ds[0]['config']['host_issue_ms_per_step']:.1f} ms, host cpu {ds[0]['config']['host_cpu_ms_per_step']:.1f} ms")
PY
}
run 1 2; run 1 1; run 2 1; run 2 2; run 3 1
cat $OUT
