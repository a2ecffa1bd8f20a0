mkdir -p gpurun_out
timeout 900 python tools/bench_part_ranking.py > gpurun_out/rank_bench.log 2>&1
tail -3 gpurun_out/rank_bench.log
