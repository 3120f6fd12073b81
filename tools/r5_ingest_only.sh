#!/bin/bash
# the ingest part of the round-5 bundle (GPU box): tests, both bench lines, the kernel trace of the text path
ROOT=$(pwd); export TMPDIR=/tmp
mkdir -p gpurun_out/r5x
python -m pytest tests/test_gpu_ingest.py tests/test_gpu_ingest_text.py -q -x 2>&1 | tail -2
python tools/bench_ingest.py --from-text --records 1000000000 --repeat 2 > gpurun_out/r5x/r5_ingest_text_1e9_bench.json 2> gpurun_out/r5x/ingest.err
python tools/bench_ingest.py --records 1000000000 > gpurun_out/r5x/r5_ingest_1e9_bench.json 2>> gpurun_out/r5x/ingest.err
(cd /tmp; rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/r5x/ing_trace -- python $ROOT/tools/bench_ingest.py --from-text --records 1000000000 --repeat 1 --no-cpu-baseline > /dev/null 2>&1)
python tools/rocprof_summary.py $(find gpurun_out/r5x/ing_trace -name "*.db" | head -1) --top 45 > gpurun_out/r5x/r5_ingest_text_1e9_kernel_stats.txt 2>&1
rm -rf gpurun_out/r5x/ing_trace
