#!/bin/bash
# QKV-epilogue test again + single-window (the reference's own usage: one 30-s window, beam 5) profile
set -o pipefail
out=gpurun_out/r03r
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 300 python3 -m pytest tests/test_gpu_gemm.py -q -k "qkv_epilogue" > $out/pytest_qkv.log 2>&1; tail -2 $out/pytest_qkv.log
python3 bench.py --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --no-overlap > $out/bench_b1.json.log 2>&1; grep -E '^\{' $out/bench_b1.json.log | cut -c1-200
rocprofv3 --kernel-trace --output-format csv -d $out/prof -o t -- python3 bench.py --batch 1 --steps 2 --warmup 1 --no-cpu-baseline --no-overlap > $out/bench_b1_prof.json.log 2>&1 || exit 1
python3 scripts/trace_summary.py $(find $out/prof -name '*kernel_trace.csv' | head -1) > $out/trace_summary_b1.txt; rm -rf $out/prof
head -24 $out/trace_summary_b1.txt | cut -c1-125
