#!/bin/bash
# SQ counters of pd_igemm_bf16 on two large plain-rows shapes (Swin-L stage 3 fc1, Swin-B stage 3 fc1): tools/debug/igemm_pmc.sh -> gpurun_out/igemm_pmc/summary.txt
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/igemm_pmc; mkdir -p $OUT
cat > /tmp/ig_drv.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from partdistillation_amd.functions import igemm
for M, K, N in ((14112, 768, 3072), (10368, 512, 2048)):
    x = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16() * 0.05; b = torch.randn(N, device="cuda").bfloat16()
    for _ in range(4): y = igemm.linear(x, w, b)
    for _ in range(4): y = torch.addmm(b, x, w.t())
torch.cuda.synchronize()
PY
i=0
for SET in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); rm -rf /tmp/igp_$i
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/igp_$i -o p -- python /tmp/ig_drv.py > $OUT/run_$i.log 2>&1
  cp /tmp/igp_$i/p_counter_collection.csv $OUT/set_$i.csv 2>/dev/null || tail -3 $OUT/run_$i.log
done
python - > $OUT/summary.txt <<'PY'
import csv, glob, collections, os, re
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/igemm_pmc")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in sorted(glob.glob(out + "/set_*.csv")):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if not ("igemm_bf16" in k or "Cijk" in k): continue
        agg[(k[:60], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in sorted(agg.items()):
    print(n)
    for c, v in sorted(d.items()): print(f"    {c:28s} {sum(v) / len(v):.4g}")
PY
cat $OUT/summary.txt
