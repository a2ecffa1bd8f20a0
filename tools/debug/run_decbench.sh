cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_declayer_gpu.py -x -q 2>&1 | tail -15
for sp in 1 2 4 8; do
  echo "=== PD_DEC_SPLIT=$sp"
  PD_DEC_SPLIT=$sp timeout 300 python -m pytest tests/test_declayer_gpu.py -x -q -k "split0" 2>&1 | tail -2
  PD_DEC_SPLIT=$sp rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/decprof_$sp -o p -- python tools/bench_declayer.py > gpurun_out/decbench_$sp.log 2>&1
  grep -v "^W2026\|^E2026" gpurun_out/decbench_$sp.log | tail -12
  f=$(find gpurun_out/decprof_$sp -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r['Name'][:70]
    if any(k in n for k in ('dec_','sgemm','add_ln','decoder_head')):
        print(f"{n:72s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:8.2f} min {float(r['MinNs'])/1e3:7.2f}")
PY
done
