#!/bin/bash
# HBM traffic of the fp32-on-bf16 GEMM kernels from the PMC counters (separate passes: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2;
# no other trace domain next to --pmc).  Usage on the GPU box: tools/pmc_gemm.sh -> gpurun_out/pmc_gemm/{FETCH_SIZE,WRITE_SIZE}.csv + summary.json
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_gemm; mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcg_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmcg_$C -o p -- python tools/pmc_gemm_driver.py > $OUT/run_$C.log 2>&1
  cp /tmp/pmcg_$C/p_counter_collection.csv $OUT/${C}.csv 2>/dev/null
done
python - <<'PY'
import csv, collections, json, os, re
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/pmc_gemm")
res = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    p = os.path.join(out, c + ".csv")
    if not os.path.exists(p):
        continue
    for r in csv.DictReader(open(p)):
        if r["Counter_Name"] != c:
            continue
        m = re.search(r"(gemm_tn_f32x3_wide<[^>]*>|gemm_tn_f32x3<[^>]*>|gemm_wgrad_f32x3_tr<[^>]*>|wgrad_tr_reduce)", r["Kernel_Name"])
        if m:
            res[m.group(1)][c].append(float(r["Counter_Value"]))
summary = {}
for k, d in res.items():
    f, w = d.get("FETCH_SIZE", []), d.get("WRITE_SIZE", [])
    summary[k] = {"launches": max(len(f), len(w)), "FETCH_SIZE_KB_avg_per_launch": sum(f) / max(len(f), 1), "WRITE_SIZE_KB_avg_per_launch": sum(w) / max(len(w), 1),
                  "hbm_bytes_corrected_avg_per_launch": (2 * sum(f) / max(len(f), 1) + sum(w) / max(len(w), 1)) * 1024}
    print(k, summary[k])
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
PY
