cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/ddptrace
PD_DDP_FORCE=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/ddptrace -o r -- python bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-categories --no-parity --skip-kernel-timing > gpurun_out/ddp_trace_bench.log 2>&1
grep '^{"metric"' gpurun_out/ddp_trace_bench.log | cut -c1-300
head -1 /tmp/ddptrace/r_kernel_trace.csv
python tools/ddp_overlap_trace.py /tmp/ddptrace/r_kernel_trace.csv | tee gpurun_out/r06_ddp_overlap_trace.txt

