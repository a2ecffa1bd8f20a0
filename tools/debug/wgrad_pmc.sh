#!/bin/bash
# SQ / LDS counters of conv_wgrad_bf16_tr on representative R50 layers: tools/debug/wgrad_pmc.sh -> gpurun_out/wgrad_pmc/summary.txt
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/wgrad_pmc; mkdir -p $OUT
cat > /tmp/wg_drv.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from partdistillation_amd.functions import conv_bf16 as OC
B = 2
for ci, co, k, hh in [(128, 128, 3, 128), (256, 64, 1, 256), (256, 256, 3, 64), (512, 128, 1, 128)]:
    x = torch.randn(B, ci, hh, hh, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, co, hh, hh, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    for _ in range(3): OC.conv_wgrad(gy, x, k, 1, k // 2)
torch.cuda.synchronize()
PY
i=0
for SET in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1)); rm -rf /tmp/wp_$i
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/wp_$i -o p -- python /tmp/wg_drv.py > $OUT/run_$i.log 2>&1
  cp /tmp/wp_$i/p_counter_collection.csv $OUT/set_$i.csv 2>/dev/null || tail -3 $OUT/run_$i.log
done
python - > $OUT/summary.txt <<'PY'
import csv, glob, collections, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/wgrad_pmc")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in sorted(glob.glob(out + "/set_*.csv")):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if "conv_wgrad_bf16_tr<" not in k: continue
        key = k[k.index("conv_wgrad"):k.index(">") + 1] + " grid " + r.get("Grid_Size", "?")
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in agg.items():
    print(n)
    for c, v in sorted(d.items()): print(f"    {c:28s} {sum(v) / len(v):.4g}")
PY
cat $OUT/summary.txt
