#!/bin/bash
# HBM-side traffic of EVERY kernel of the training step: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, only --kernel-trace next
# to them) over a short bench.py run; per-kernel averages of the last steps -> gpurun_out/pmc_step/summary.csv
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_step; mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcs_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmcs_$C -o p -- python bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-categories --no-parity --skip-kernel-timing > $OUT/run_$C.log 2>&1
  cp /tmp/pmcs_$C/p_counter_collection.csv $OUT/${C}.csv 2>/dev/null
done
python - <<'PY'
import csv, collections, os, re
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/pmc_step")
agg = collections.defaultdict(lambda: {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
steps = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    p = os.path.join(out, c + ".csv")
    if not os.path.exists(p):
        continue
    rows = [r for r in csv.DictReader(open(p)) if r["Counter_Name"] == c]
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
    rows = rows[len(rows) // 2:]                     # the second half of the dispatches: the timed steps (+ whatever bench.py runs after them)
    steps[c] = max(1, sum(1 for r in rows if "uncertain_points" in r["Kernel_Name"]))      # one launch per training step
    for r in rows:
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        k = re.sub(r"\((?!anonymous).*", "", k)[:110]
        agg[k][c][0] += float(r["Counter_Value"]); agg[k][c][1] += 1
with open(os.path.join(out, "summary.csv"), "w") as f:
    f.write("kernel,launches_per_step,fetch_MB_x2_per_launch,write_MB_per_launch,hbm_MB_per_launch,hbm_MB_per_step\n")
    tot, ns = 0.0, max(steps.values()) if steps else 3
    for k, d in sorted(agg.items(), key=lambda kv: -(2 * kv[1]["FETCH_SIZE"][0] + kv[1]["WRITE_SIZE"][0])):
        n = max(d["FETCH_SIZE"][1], d["WRITE_SIZE"][1], 1)
        fe, wr = 2 * d["FETCH_SIZE"][0] / 1024 / n, d["WRITE_SIZE"][0] / 1024 / n
        tot += (fe + wr) * n / ns
        f.write(f"\"{k}\",{n / ns:.1f},{fe:.1f},{wr:.1f},{fe + wr:.1f},{(fe + wr) * n / ns:.1f}\n")
print(open(os.path.join(out, "summary.csv")).read()[:6000])
print("total MB per step:", round(tot))
PY
