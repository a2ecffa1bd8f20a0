"""Merge the per-test parity records the full-size GPU tests leave in gpurun_out/parity/*.json into ONE committed file:
    python tools/collect_parity.py profiles/r04_parity.json"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "parity.json")
recs = {}
for p in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "parity", "*.json"))):
    r = json.load(open(p))
    recs[r.pop("test")] = r
json.dump({"source": "tests/test_product_gpu.py full-size parity tests (HIP step vs CPU oracle on identical inputs), written by _record_parity",
           "records": recs}, open(out, "w"), indent=1, sort_keys=True)
print(out, list(recs))
