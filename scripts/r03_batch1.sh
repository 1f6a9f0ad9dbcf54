#!/bin/bash
# round-3 batch 1: GEMM epilogue A/B (old vs new harness binaries), self-attention non-temporal A/B, correctness of the new epilogues
set -o pipefail
out=gpurun_out/r03b
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 600 python3 -m pytest tests/test_gpu_gemm.py tests/test_gpu_attention.py tests/test_gpu_model.py tests/test_gpu_decode.py tests/test_gpu_wide.py -x -q -k "not 480_rows" > $out/pytest_subset.log 2>&1 || { tail -30 $out/pytest_subset.log; exit 1; }
tail -3 $out/pytest_subset.log
for rep in 1 2; do
  for v in old new; do
    echo "== $v enc 307 (rep $rep)" >> $out/gemm_ab.log
    timeout -k 10 120 scripts/micro/gemm_f16_bench_$v enc 307 >> $out/gemm_ab.log 2>&1 || exit 1
    echo "== $v dec 1535 (rep $rep)" >> $out/gemm_ab.log
    timeout -k 10 120 scripts/micro/gemm_f16_bench_$v dec 1535 >> $out/gemm_ab.log 2>&1 || exit 1
  done
done
echo "gemm ab done"
for nt in 0 1; do
  WRS_SELF_NT=$nt rocprofv3 --kernel-trace --output-format csv -d $out/self_nt$nt -o t -- python3 -m pytest tests/test_gpu_attention.py -q -k "self_attention" > $out/self_nt$nt.log 2>&1 || { tail -20 $out/self_nt$nt.log; exit 1; }
  python3 scripts/trace_summary.py $(find $out/self_nt$nt -name '*kernel_trace.csv' | head -1) > $out/self_nt${nt}_summary.txt
  rm -rf $out/self_nt$nt
done
echo "self nt done"
