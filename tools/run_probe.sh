mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_window_attention_gpu.py -x -q > gpurun_out/wattn_test.log 2>&1
tail -3 gpurun_out/wattn_test.log
timeout 600 python tools/bench_window_attention.py > gpurun_out/wattn_bench.log 2>&1
grep grid gpurun_out/wattn_bench.log
PD_CONFIG=swinl timeout 600 python tools/bench_config3.py 1280 6 > gpurun_out/config5_wattn.log 2>&1
tail -1 gpurun_out/config5_wattn.log
timeout 600 python tools/bench_config3.py 1024 8 > gpurun_out/config3_wattn.log 2>&1
tail -1 gpurun_out/config3_wattn.log
