#!/bin/bash
# Host-side cost of N training processes on ONE host: N independent bench.py processes on the one GPU of a development
# box (no gradient exchange: gloo's all-reduce of device tensors stages 176 MB per rank through host memory and TCP loopback — 96 ms per
# step at N = 2 — and says nothing about RCCL).  The processes share the GPU, so the step time grows ~N x; what must NOT grow is a process's
# host CPU per step (config.host_cpu_ms_per_step): eight ranks will share one host's cores.
# Usage (GPU box, repo root): tools/shared_gpu_ranks.sh [size]   -> gpurun_out/shared_ranks.txt
SIZE=${1:-512}
OUT=gpurun_out/shared_ranks.txt; mkdir -p gpurun_out; : > $OUT
echo "host cores: $(nproc)" >> $OUT
for N in 1 2 4 8; do
  pids=""
  for r in $(seq 1 $N); do
    python bench.py --gpus 1 --steps 10 --warmup 4 --size $SIZE --no-cpu-baseline --no-categories --no-parity --skip-kernel-timing 2>/dev/null | grep '^{"metric' > /tmp/sr_$r.json &
    pids="$pids $!"
  done
  for p in $pids; do wait $p; done
  python - $N >> $OUT <<'PY'
import json, sys
n = int(sys.argv[1])
ds = [json.load(open(f'/tmp/sr_{r}.json')) for r in range(1, n + 1)]
ms = [d['ms_per_step'] for d in ds]; cpu = [d['config']['host_cpu_ms_per_step'] for d in ds]
print(f"{n} processes on one GPU: {min(ms):.1f} .. {max(ms):.1f} ms per step each ({sum(d['value'] for d in ds):.1f} images/s together), host CPU per step and process {min(cpu):.1f} .. {max(cpu):.1f} ms")
PY
done
cat $OUT
