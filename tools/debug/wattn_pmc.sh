#!/bin/bash
# HBM traffic + SQ counters of the window-attention kernels at Swin-B's stage 1 / stage 3 shapes: tools/debug/wattn_pmc.sh -> gpurun_out/wattn_pmc/summary.txt
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/wattn_pmc; mkdir -p $OUT
cat > /tmp/wa_drv.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from partdistillation_amd.functions import window_attention as wa
for side, heads in ((264, 4), (72, 16)):
    nW = (side // 12) ** 2
    B_, C = 2 * nW, heads * 32
    qkv = torch.randn(B_, 144, 3 * C, device="cuda").bfloat16()
    table = torch.randn(529, heads, device="cuda") * 0.1
    go = torch.randn(B_, 144, C, device="cuda").bfloat16()
    for _ in range(3):
        out, lse = wa.fwd_raw(qkv, table, None, 32 ** -0.5, nW)
        wa.bwd_raw(qkv, table, None, out, go, lse, 32 ** -0.5, nW)
torch.cuda.synchronize()
PY
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1)); rm -rf /tmp/wap_$i
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/wap_$i -o p -- python /tmp/wa_drv.py > $OUT/run_$i.log 2>&1
  cp /tmp/wap_$i/p_counter_collection.csv $OUT/set_$i.csv 2>/dev/null || tail -3 $OUT/run_$i.log
done
python - > $OUT/summary.txt <<'PY'
import csv, glob, collections, os, re
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/wattn_pmc")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in sorted(glob.glob(out + "/set_*.csv")):
    for r in csv.DictReader(open(p)):
        m = re.search(r"(wattn_\w+)", r["Kernel_Name"])
        if not m: continue
        agg[(m.group(1), r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in sorted(agg.items()):
    print(n)
    for c, v in sorted(d.items()): print(f"    {c:28s} {sum(v) / len(v):.4g}")
PY
cat $OUT/summary.txt
