#!/bin/bash
# round-3 batch 7: full GPU suite on the two-library build (tests load libwrs_mi355x_debug.so), then a sequential pass of the
# product library under rocprofv3 (filter kernel with three iterations requested ahead) and the default bench
set -o pipefail
out=gpurun_out/r03h
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; tail -6 $out/pytest_gpu.log; grep -E "config 1|config1" $out/pytest_gpu.log | head -5
rocprofv3 --kernel-trace --output-format csv -d $out/prof -o t -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap --no-shard-check > $out/bench_seq.json.log 2>&1 || { tail -5 $out/bench_seq.json.log; exit 1; }
python3 scripts/trace_summary.py $(find $out/prof -name '*kernel_trace.csv' | head -1) > $out/trace_summary_seq.txt; rm -rf $out/prof
head -16 $out/trace_summary_seq.txt | cut -c1-125
python3 bench.py --steps 3 --warmup 1 > $out/bench_default.json.log 2>&1 || { tail -5 $out/bench_default.json.log; exit 1; }
grep -E '^\{' $out/bench_default.json.log | cut -c1-300
