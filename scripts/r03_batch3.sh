#!/bin/bash
# round-3 batch 3: cache-policy bits of direct-to-LDS loads (micro + GEMM harness), in-situ A/B of the round-start library
# against the current one (sequential pass under rocprofv3, same box), self-attention non-temporal loads in situ
set -o pipefail
out=gpurun_out/r03d
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for a in 0 1 2 16 17 18; do
  for cfg in "2 2 0" "1 3 0"; do
    echo -n "aux $a: " >> $out/l2_to_lds_aux.log
    timeout -k 10 60 scripts/micro/l2_to_lds_aux$a $cfg >> $out/l2_to_lds_aux.log 2>&1 || exit 1
  done
done
echo "l2 aux done"
for v in new a16_w16 a16_w0 a0_w16 a2_w2 a0_w2 new; do
  echo "== $v dec 1535" >> $out/gemm_aux.log
  timeout -k 10 120 scripts/micro/gemm_f16_bench_$v dec 1535 >> $out/gemm_aux.log 2>&1 || exit 1
  echo "== $v enc 307" >> $out/gemm_aux.log
  timeout -k 10 120 scripts/micro/gemm_f16_bench_$v enc 307 >> $out/gemm_aux.log 2>&1 || exit 1
done
echo "gemm aux done"
prof() {  # $1 tag
  rocprofv3 --kernel-trace --output-format csv -d $out/prof_$1 -o t -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap --no-shard-check > $out/bench_$1.json.log 2>&1 || { tail -5 $out/bench_$1.json.log; return 1; }
  python3 scripts/trace_summary.py $(find $out/prof_$1 -name '*kernel_trace.csv' | head -1) > $out/trace_summary_$1.txt
  rm -rf $out/prof_$1
  echo "prof $1 done"
}
prof cur || exit 1
WRS_SELF_NT=0 prof selfplain || exit 1
cp whisper_rs_amd/libwrs_mi355x.so $out/../lib_keep.so && cp scripts/micro/libwrs_r03start.so whisper_rs_amd/libwrs_mi355x.so
prof start || exit 1
cp $out/../lib_keep.so whisper_rs_amd/libwrs_mi355x.so; rm -f $out/../lib_keep.so
