#!/bin/bash
# filter kernel with the wave-level K-th-best bound: bit-exact filter tests + decode tests, then the kernel's average in a sequential pass
set -o pipefail
out=gpurun_out/r03z3
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 600 python3 -m pytest tests/test_gpu_filters.py tests/test_gpu_decode.py tests/test_gpu_golden.py tests/test_gpu_configs.py tests/test_gpu_parity_f32.py -x -q > $out/pytest_subset.log 2>&1 || { tail -30 $out/pytest_subset.log; exit 1; }
tail -2 $out/pytest_subset.log
rocprofv3 --kernel-trace --output-format csv -d $out/prof -o t -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap --no-shard-check > $out/bench_seq.json.log 2>&1 || exit 1
python3 scripts/trace_summary.py $(find $out/prof -name '*kernel_trace.csv' | head -1) > $out/trace_summary_seq.txt; rm -rf $out/prof
grep -E "filter_topk|attn_mfma_kernel|gemm_f16_kernel<true>.*1247232" $out/trace_summary_seq.txt | cut -c1-125
