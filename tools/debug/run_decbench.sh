cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_declayer_gpu.py -x -q 2>&1 | tail -3
for sp in 4 8; do
PD_DEC_SPLIT=$sp timeout 300 python -m pytest tests/test_declayer_gpu.py -x -q 2>&1 | tail -1
PD_DEC_SPLIT=$sp rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/decprof_$sp -o p -- python tools/bench_declayer.py > gpurun_out/decbench.log 2>&1
f=$(find gpurun_out/decprof_$sp -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r['Name'][:70]
    if any(k in n for k in ('dec_',)):
        print(f"{n:72s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:8.2f} min {float(r['MinNs'])/1e3:7.2f}")
PY
done
bash tools/ab_bench.sh "one launch|PD_DEC_SPLIT=1" "split 4|PD_DEC_SPLIT=4" "split 8|PD_DEC_SPLIT=8" "split 2|PD_DEC_SPLIT=2" "one launch|PD_DEC_SPLIT=1" "split 4|PD_DEC_SPLIT=4" "split 8|PD_DEC_SPLIT=8"
