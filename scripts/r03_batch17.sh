#!/bin/bash
# mid-size batches: which self-attention form from which size?  (test build: switches honoured)
out=gpurun_out/r03x
mkdir -p $out
run() {  # $1 batch, rest: env assignments
  b=$1; shift
  env "$@" python3 bench.py --batch $b --steps 3 --warmup 1 --no-cpu-baseline --no-overlap --debug-lib > $out/run.log 2>&1
  grep -E '^\{' $out/run.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch', $b, '$*', d['value'], d['ms_per_step'])"
}
for b in 8 16 32; do
  run $b WRS_X=0
  run $b WRS_SELF_MFMA_MIN_AUDIO=8
  run $b WRS_SELF_PSPLIT_ROWS=400
done
run 64 WRS_X=0
run 64 WRS_SELF_MFMA_MIN_AUDIO=100
run 64 WRS_SELF_MFMA_MIN_AUDIO=100 WRS_SELF_PSPLIT_ROWS=400
