#!/bin/bash
# encoder MLP-2 as 2 / 4 launches over slices of K (shorter-lived tiles on the side stream): overlapped bench A/B, interleaved, one box
out=gpurun_out/r03ms
mkdir -p $out
WRS_ENC_MLP2_SPLIT=2 timeout -k 10 400 python3 -m pytest tests/test_gpu_wide.py -x -q -k "encoder" > $out/pytest_wide_enc.log 2>&1; tail -2 $out/pytest_wide_enc.log
run() {
  env "$@" python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --debug-lib > $out/run.log 2>&1
  grep -E '^\{' $out/run.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['last_step_split_ms'])"
}
for rep in 1 2; do
  run WRS_ENC_MLP2_SPLIT=1
  run WRS_ENC_MLP2_SPLIT=2
  run WRS_ENC_MLP2_SPLIT=4
done
