#!/bin/bash
# HBM traffic of the fp32-on-bf16 GEMM kernels from the PMC counters (separate passes: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2;
# no other trace domain next to --pmc).  Usage on the GPU box: tools/pmc_gemm.sh -> gpurun_out/pmc_gemm/{FETCH_SIZE,WRITE_SIZE}.csv + summary.json
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_gemm; mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcg_$C
  PMC_LABELS=/tmp/pmc_labels.json rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmcg_$C -o p -- python tools/pmc_gemm_driver.py > $OUT/run_$C.log 2>&1
  cp /tmp/pmcg_$C/p_counter_collection.csv $OUT/${C}.csv 2>/dev/null
done
python - <<'PY'
import csv, collections, json, os, re
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/pmc_gemm")
lab = json.load(open("/tmp/pmc_labels.json"))
res = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    p = os.path.join(out, c + ".csv")
    if not os.path.exists(p):
        continue
    rows = [r for r in csv.DictReader(open(p)) if r["Counter_Name"] == c]
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
    fwd = [r for r in rows if re.search(r"gemm_tn_f16x2<|gemm_rows_f16x2_k256<|gemm_kpc_f16x2<|gemm_kres_f16x2<", r["Kernel_Name"])]   # (whichever kernel of the family a shape takes)
    assert len(fwd) == len(lab["fwd"]), (len(fwd), len(lab["fwd"]))
    for r, l in zip(fwd, lab["fwd"]):                       # dispatch order = the order the driver issued them in
        res[l][c].append(float(r["Counter_Value"]))
    g = [r for r in rows if re.search(r"gemm_wgrad_f32x3_tr_grouped|wgrad_tr_reduce_grouped|gemm_wgrad_f16x2_wide_grouped|wgrad_h2w_reduce_grouped", r["Kernel_Name"])]
    for i in range(0, len(g), 4):                           # the layer's five weight gradients: wide group + narrow group, each with its reduce
        # bench.py times the step's TWO grouped launches (6 layers' problems, wide and narrow tiles) and reports the average of the two
        res["gemm_wgrad_f16x2_wide_grouped / gemm_wgrad_f32x3_tr_grouped<f16x2> (+ their reduces)"][c].append(0.5 * sum(float(r["Counter_Value"]) for r in g[i:i + 4]))
summary = {}
for k, d in res.items():
    f, w = d.get("FETCH_SIZE", []), d.get("WRITE_SIZE", [])
    summary[k] = {"launches": max(len(f), len(w)), "FETCH_SIZE_KB_avg_per_launch": sum(f) / max(len(f), 1), "WRITE_SIZE_KB_avg_per_launch": sum(w) / max(len(w), 1),
                  "hbm_bytes_corrected_avg_per_launch": (2 * sum(f) / max(len(f), 1) + sum(w) / max(len(w), 1)) * 1024,
                  "note": "FETCH_SIZE x 2 (gfx950 counts 64-byte units as 32: MI355X_MICROARCH.md) + WRITE_SIZE, KB -> bytes; averages over the mix one encoder layer issues (tools/pmc_gemm_driver.py)"}
    print(k, summary[k])
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
PY
