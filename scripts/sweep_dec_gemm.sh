#!/bin/bash
# sweep: tiled (thr=1) vs split-K latency kernel (thr=100000) for the decode GEMMs at several batch sizes
for b in "$@"; do
  for thr in 1 100000; do
    WRS_DEC_BIG_GEMM_ROWS=$thr timeout -k 10 500 python bench.py --batch $b --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b_x.log 2>&1
    tail -1 gpurun_out/b_x.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch', $b, 'thr', $thr, d['value'], d['last_step_split_ms'])"
  done
done
