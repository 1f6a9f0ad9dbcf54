#!/bin/bash
# overlapped against sequential passes on one box (no profiler), interleaved
out=gpurun_out/r03u
mkdir -p $out
for rep in 1 2; do
  python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/ov_$rep.log 2>&1; grep -E '^\{' $out/ov_$rep.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('overlapped', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['last_step_split_ms'])"
  python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap > $out/seq_$rep.log 2>&1; grep -E '^\{' $out/seq_$rep.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('sequential', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['last_step_split_ms'])"
done
