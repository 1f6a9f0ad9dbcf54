#!/bin/bash
# round-3 batch 9: work-stealing self-attention (units handed out by a counter to a grid of resident waves): tests, kernel test and in situ A/B
set -o pipefail
out=gpurun_out/r03k
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 600 python3 -m pytest tests/test_gpu_attention.py tests/test_gpu_filters.py tests/test_gpu_decode.py tests/test_gpu_model.py tests/test_gpu_wide.py tests/test_gpu_golden.py -x -q > $out/pytest_subset.log 2>&1 || { tail -30 $out/pytest_subset.log; exit 1; }
tail -2 $out/pytest_subset.log
for f in 1 0; do
  WRS_SELF_STEAL=$f rocprofv3 --kernel-trace --output-format csv -d $out/self_steal$f -o t -- python3 -m pytest tests/test_gpu_attention.py -q -k "self_attention" > $out/self_steal$f.log 2>&1 || { tail -20 $out/self_steal$f.log; exit 1; }
  python3 scripts/trace_summary.py $(find $out/self_steal$f -name '*kernel_trace.csv' | head -1) > $out/self_steal${f}_summary.txt
  rm -rf $out/self_steal$f
  echo "steal $f:"; grep -E "attn_self_mfma" $out/self_steal${f}_summary.txt | head -2
done
prof() {  # $1 tag
  rocprofv3 --kernel-trace --output-format csv -d $out/prof_$1 -o t -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap --no-shard-check --debug-lib > $out/bench_$1.json.log 2>&1 || { tail -5 $out/bench_$1.json.log; return 1; }
  python3 scripts/trace_summary.py $(find $out/prof_$1 -name '*kernel_trace.csv' | head -1) > $out/trace_summary_$1.txt
  rm -rf $out/prof_$1
  echo "prof $1 done"; grep -E "attn_self_mfma|attn_mfma|pack_kv" $out/trace_summary_$1.txt | head -4
}
WRS_SELF_STEAL=1 prof steal1 || exit 1
WRS_SELF_STEAL=0 prof steal0 || exit 1
