mkdir -p gpurun_out
timeout 600 python tools/bench_config3.py 1024 8 > gpurun_out/config3_wg.log 2>&1
tail -1 gpurun_out/config3_wg.log
PD_CONFIG=swinl timeout 600 python tools/bench_config3.py 1280 6 > gpurun_out/config5_wg.log 2>&1
tail -1 gpurun_out/config5_wg.log
