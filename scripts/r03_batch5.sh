#!/bin/bash
# bisect the config-1 divergence: current library, IEEE reciprocal in GELU, plain (temporal) loads, both
out=gpurun_out/r03f
mkdir -p $out
cp whisper_rs_amd/libwrs_mi355x.so $out/../lib_keep.so
for tag in cur ieee nt0 both; do
  if [ $tag != cur ]; then cp scripts/micro/libwrs_$tag.so whisper_rs_amd/libwrs_mi355x.so; fi
  timeout -k 10 300 python3 -m pytest tests/test_gpu_configs.py -q -k "config1 or config2" > $out/config1_$tag.log 2>&1
  echo "== $tag: $(tail -1 $out/config1_$tag.log)"
done
cp $out/../lib_keep.so whisper_rs_amd/libwrs_mi355x.so; rm -f $out/../lib_keep.so
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $out/pytest_gpu_all.log 2>&1; tail -12 $out/pytest_gpu_all.log
