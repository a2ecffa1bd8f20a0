mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_part_ranking.py tests/test_product_gpu.py tests/test_propgen_gpu.py -x -q -m gpu > gpurun_out/save_test.log 2>&1
tail -15 gpurun_out/save_test.log
