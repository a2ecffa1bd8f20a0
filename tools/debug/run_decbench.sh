cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for pin in 0 1; do
  PD_DEC_XCD_PIN=$pin timeout 300 python -m pytest tests/test_declayer_gpu.py -x -q 2>&1 | tail -2
  PD_DEC_XCD_PIN=$pin rocprofv3 --kernel-trace --stats -d gpurun_out/decprof_$pin -o p -- python tools/bench_declayer.py > gpurun_out/decbench_$pin.log 2>&1
  tail -12 gpurun_out/decbench_$pin.log
  f=$(find gpurun_out/decprof_$pin -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r['Name'][:60]
    if any(k in n for k in ('dec_','sgemm','add_ln','decoder_head')):
        print(f"{n:62s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:8.2f} min {float(r['MinNs'])/1e3:7.2f}")
PY
done
