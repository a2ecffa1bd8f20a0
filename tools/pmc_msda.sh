#!/bin/bash
# HBM traffic of the MSDA kernels from the PMC counters (separate passes: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2).
# Usage on the GPU box: tools/pmc_msda.sh  -> gpurun_out/pmc_msda/{fetch,write}_*.csv + summary.txt
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_msda; mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$C -o p -- python tools/bench_msda.py --iters 10 --px 0.5 --fused ${FUSED:-1} --ablate ${ABLATE:-0} --variant ${VARIANT:-0} > /dev/null 2>&1
  cp /tmp/pmc_$C/p_counter_collection.csv $OUT/${C}.csv 2>/dev/null
done
python - <<'PY'
import csv, collections, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/pmc_msda")
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    p = os.path.join(out, c + ".csv")
    if not os.path.exists(p):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if "msda" in r["Kernel_Name"] and r["Counter_Name"] == c:
            import re as _re; agg[(_re.search(r"(msda_\w+)", r["Kernel_Name"]) or [None, "?"])[1] if _re.search(r"(msda_\w+)", r["Kernel_Name"]) else "?"].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        res[k][c] = sum(v) / len(v)
with open(os.path.join(out, "summary.txt"), "w") as f:
    for k, d in res.items():
        line = f"{k}: " + ", ".join(f"{c} avg/launch = {v:.1f} (counter units of KB -> {v/1024:.1f} MB)" for c, v in d.items())
        print(line); f.write(line + "\n")
PY
