mkdir -p gpurun_out
timeout 900 python tools/bench_input_pipeline.py > gpurun_out/input_bench.log 2>&1
tail -3 gpurun_out/input_bench.log
