#!/bin/bash
# round-3 batch 4: 128x64 ring GEMM without split-K (fused residual epilogue) + LayerNorm-only finish, then the full GPU suite
set -o pipefail
out=gpurun_out/r03e
mkdir -p $out
for rep in 1 2; do
  timeout -k 10 120 scripts/micro/gemm_f16_bench_new dec 1535 >> $out/gemm_ring64.log 2>&1 || exit 1
  timeout -k 10 60 scripts/micro/finish_bench 1535 >> $out/finish_bench.log 2>&1 || exit 1
done
cat $out/finish_bench.log; grep -E "d_DxD|d_mlp2" $out/gemm_ring64.log
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; tail -4 $out/pytest_gpu.log
