mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1
tail -4 gpurun_out/gpu_tests.log
timeout 900 python bench.py > gpurun_out/bench_v4.log 2>&1
grep '^{"metric"' gpurun_out/bench_v4.log | cut -c1-600
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/prof_c3
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -o r -- python tools/bench_config3.py 1024 6 > gpurun_out/config3_rocprof.log 2>&1
tail -1 gpurun_out/config3_rocprof.log | cut -c1-300
ls /tmp/prof_c3 | head
cp /tmp/prof_c3/r_kernel_stats.csv gpurun_out/config3_kernel_stats.csv 2>/dev/null || find /tmp/prof_c3 -name "*kernel_stats.csv" -exec cp {} gpurun_out/config3_kernel_stats.csv \;
head -12 gpurun_out/config3_kernel_stats.csv | cut -c1-200
