#!/bin/bash
# round-3 batch 2: default bench with the new kernels, then per-kernel A/B under rocprofv3 (sequential pass):
#   lib (streaming loads on, XCD row order) / WRS_FINISH_PLAIN_ROWS=1 / scripts/micro/libwrs_nt0.so (-DWRS_NT_STREAMS=0)
set -o pipefail
out=gpurun_out/r03c
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_default.json.log 2>&1 || { tail -5 $out/bench_default.json.log; exit 1; }
tail -1 $out/bench_default.json.log | cut -c1-400
prof() {  # $1 tag
  rocprofv3 --kernel-trace --output-format csv -d $out/prof_$1 -o t -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap --no-shard-check > $out/bench_$1.json.log 2>&1 || { tail -5 $out/bench_$1.json.log; return 1; }
  python3 scripts/trace_summary.py $(find $out/prof_$1 -name '*kernel_trace.csv' | head -1) > $out/trace_summary_$1.txt
  rm -rf $out/prof_$1
  echo "prof $1 done"
}
prof new || exit 1
WRS_FINISH_PLAIN_ROWS=1 prof plainrows || exit 1
cp whisper_rs_amd/libwrs_mi355x.so $out/../lib_keep.so && cp scripts/micro/libwrs_nt0.so whisper_rs_amd/libwrs_mi355x.so
prof nt0 || exit 1
cp $out/../lib_keep.so whisper_rs_amd/libwrs_mi355x.so; rm -f $out/../lib_keep.so
