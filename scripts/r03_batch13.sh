#!/bin/bash
# 16-wave cross-attention at few workgroups: correctness (small-batch decode tests) + single-window bench and kernel average
set -o pipefail
out=gpurun_out/r03s
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 600 python3 -m pytest tests/test_gpu_decode.py tests/test_gpu_model.py tests/test_gpu_golden.py tests/test_gpu_transcribe.py tests/test_gpu_attention.py -x -q > $out/pytest_subset.log 2>&1 || { tail -30 $out/pytest_subset.log; exit 1; }
tail -2 $out/pytest_subset.log
python3 bench.py --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --no-overlap > $out/bench_b1.json.log 2>&1; grep -E '^\{' $out/bench_b1.json.log | cut -c1-200
rocprofv3 --kernel-trace --output-format csv -d $out/prof -o t -- python3 bench.py --batch 1 --steps 2 --warmup 1 --no-cpu-baseline --no-overlap > $out/bench_b1_prof.json.log 2>&1 || exit 1
python3 scripts/trace_summary.py $(find $out/prof -name '*kernel_trace.csv' | head -1) > $out/trace_summary_b1.txt; rm -rf $out/prof
head -8 $out/trace_summary_b1.txt | cut -c1-125
python3 bench.py --batch 3 --steps 4 --warmup 1 --no-cpu-baseline --no-overlap > $out/bench_b3.json.log 2>&1; grep -E '^\{' $out/bench_b3.json.log | cut -c1-200
