#!/bin/bash
# SQ / LDS counters of the three 256 <- 1024 kernels (tiled, resident accumulators, producer / consumer): tools/debug/kres_pmc.sh -> gpurun_out/kres_pmc/
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/kres_pmc; mkdir -p $OUT
cat > /tmp/kres_drv.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from partdistillation_amd import lib; L = lib.load()
from partdistillation_amd.functions import gemm
M, N, K = 43008, 256, 1024
a, w = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") * K ** -0.5
aa, wa = gemm.row_amax(a), gemm.row_amax(w)
for v in (80, 91, 92):
    L.pd_debug_set(b"f16x2_tile", v)
    for _ in range(3): gemm.gemm_tn_h2(a, w, None, a_amax=aa, b_amax=wa)
torch.cuda.synchronize()
PY
i=0
for SET in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"; do
  i=$((i+1)); rm -rf /tmp/kp_$i
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/kp_$i -o p -- python /tmp/kres_drv.py > $OUT/run_$i.log 2>&1
  cp /tmp/kp_$i/p_counter_collection.csv $OUT/set_$i.csv 2>/dev/null || tail -3 $OUT/run_$i.log
done
python - <<'PY'
import csv, glob, collections, os, re
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/kres_pmc")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in sorted(glob.glob(out + "/set_*.csv")):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if "gemm_tn_f16x2<" in k: n = "tiled"
        elif "gemm_kres" in k: n = "kres"
        elif "gemm_kpc" in k: n = "kpc"
        else: continue
        agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in agg.items():
    print(n, {c: f"{sum(v) / len(v):.3g}" for c, v in sorted(d.items())})
PY
