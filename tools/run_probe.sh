mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_swin_stage_gpu.py tests/test_product_gpu.py -x -q > gpurun_out/stage_test.log 2>&1
tail -3 gpurun_out/stage_test.log
timeout 600 python tools/bench_config3.py 1024 8 > gpurun_out/config3_stage.log 2>&1
tail -1 gpurun_out/config3_stage.log
PD_CONFIG=swinl timeout 600 python tools/bench_config3.py 1280 6 > gpurun_out/config5_stage.log 2>&1
tail -1 gpurun_out/config5_stage.log
