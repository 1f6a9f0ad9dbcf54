#!/bin/bash
# K / V projections packed by the GEMM epilogue (WRS_FUSED_PACK=1, test build): wide-model parity tests, then sequential passes with and without it
set -o pipefail
out=gpurun_out/r03fp
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
WRS_FUSED_PACK=1 timeout -k 10 800 python3 -m pytest tests/test_gpu_wide.py tests/test_gpu_gemm.py -x -q -k "not slab_path" > $out/pytest_wide_fused.log 2>&1 || { tail -30 $out/pytest_wide_fused.log; exit 1; }
tail -2 $out/pytest_wide_fused.log
prof() {  # $1 tag
  rocprofv3 --kernel-trace --output-format csv -d $out/prof_$1 -o t -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap --no-shard-check --debug-lib > $out/bench_$1.json.log 2>&1 || { tail -5 $out/bench_$1.json.log; return 1; }
  python3 scripts/trace_summary.py $(find $out/prof_$1 -name '*kernel_trace.csv' | head -1) > $out/trace_summary_$1.txt
  rm -rf $out/prof_$1
  echo "prof $1:"; grep -E "pp2|pack_kv|attn_enc|attn_mfma_kernel" $out/trace_summary_$1.txt | cut -c1-125
  grep -E '^\{' $out/bench_$1.json.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['last_step_split_ms'])"
}
WRS_FUSED_PACK=1 prof fused || exit 1
WRS_FUSED_PACK=0 prof unfused || exit 1
