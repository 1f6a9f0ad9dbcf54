#!/bin/bash
# round-3 batch 6: config-1 test through the near-tie path; fragment-order self-attention (WRS_SELF_FRAG=1): kernel test + in situ
set -o pipefail
out=gpurun_out/r03g
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 300 python3 -m pytest tests/test_gpu_configs.py -q -s -k "config1" > $out/config1.log 2>&1; echo "config1: $(tail -1 $out/config1.log)"; grep -E "config 1|parity\]" $out/config1.log | cut -c1-300
for f in 0 1; do
  WRS_SELF_FRAG=$f rocprofv3 --kernel-trace --output-format csv -d $out/self_frag$f -o t -- python3 -m pytest tests/test_gpu_attention.py tests/test_gpu_wide.py -q -k "self_attention or 480_rows" > $out/self_frag$f.log 2>&1 || { tail -20 $out/self_frag$f.log; exit 1; }
  python3 scripts/trace_summary.py $(find $out/self_frag$f -name '*kernel_trace.csv' | head -1) > $out/self_frag${f}_summary.txt
  rm -rf $out/self_frag$f
  echo "frag $f: $(tail -1 $out/self_frag$f.log)"; grep -E "attn_self_mfma" $out/self_frag${f}_summary.txt | head -3
done
prof() {  # $1 tag
  rocprofv3 --kernel-trace --output-format csv -d $out/prof_$1 -o t -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap --no-shard-check --debug-lib > $out/bench_$1.json.log 2>&1 || { tail -5 $out/bench_$1.json.log; return 1; }
  python3 scripts/trace_summary.py $(find $out/prof_$1 -name '*kernel_trace.csv' | head -1) > $out/trace_summary_$1.txt
  rm -rf $out/prof_$1
  echo "prof $1 done"; grep -E "attn_self_mfma|attn_mfma" $out/trace_summary_$1.txt | head -3
}
WRS_SELF_FRAG=0 prof frag0 || exit 1
WRS_SELF_FRAG=1 prof frag1 || exit 1
