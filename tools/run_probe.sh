mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fp8_gpu.py -x -q > gpurun_out/fp8_test.log 2>&1
tail -5 gpurun_out/fp8_test.log
