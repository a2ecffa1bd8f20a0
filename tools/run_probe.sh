mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_smallgemm_gpu.py tests/test_swin_stage_gpu.py -x -q > gpurun_out/wg_test.log 2>&1
tail -5 gpurun_out/wg_test.log
timeout 600 python tools/bench_wgrad_split.py > gpurun_out/wg_bench.log 2>&1
grep "M=" gpurun_out/wg_bench.log
timeout 600 python tools/bench_config3.py 1024 8 > gpurun_out/config3_wg.log 2>&1
tail -1 gpurun_out/config3_wg.log
PD_CONFIG=swinl timeout 600 python tools/bench_config3.py 1280 6 > gpurun_out/config5_wg.log 2>&1
tail -1 gpurun_out/config5_wg.log
