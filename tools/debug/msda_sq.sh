#!/bin/bash
# SQ / LDS counters of the fused MSDA kernels at config-2 geometry: tools/debug/msda_sq.sh -> gpurun_out/msda_sq/summary.txt
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/msda_sq; mkdir -p $OUT
i=0
for SET in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_INSTS_VMEM_RD"; do
  i=$((i+1)); rm -rf /tmp/ms_$i
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/ms_$i -o p -- python tools/bench_msda.py --iters 4 --px 0.5 --fused 1 > $OUT/run_$i.log 2>&1
  cp /tmp/ms_$i/p_counter_collection.csv $OUT/set_$i.csv 2>/dev/null || tail -3 $OUT/run_$i.log
done
python - > $OUT/summary.txt <<'PY'
import csv, glob, collections, os, re
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/msda_sq")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in sorted(glob.glob(out + "/set_*.csv")):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        m = re.search(r"(msda_\w+)", k)
        if not m: continue
        agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in agg.items():
    print(n)
    for c, v in sorted(d.items()): print(f"    {c:28s} {sum(v) / len(v):.4g}")
PY
cat $OUT/summary.txt
