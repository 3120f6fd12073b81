ROOT=$(pwd); export TMPDIR=/tmp
mkdir -p gpurun_out/r5x
python -m pytest tests/test_gpu_topn.py -q -x 2>&1 | tail -2
python tools/bench_topn.py > gpurun_out/r5x/r5_topn_1M_bench.json 2> gpurun_out/r5x/topn.err
(cd /tmp; rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/r5x/topn_trace -- python $ROOT/tools/bench_topn.py --no-cpu-baseline > /dev/null 2>&1)
python tools/rocprof_summary.py $(find gpurun_out/r5x/topn_trace -name "*.db" | head -1) --top 25 > gpurun_out/r5x/r5_topn_1M_kernel_stats.txt 2>&1
rm -rf gpurun_out/r5x/topn_trace
