#!/bin/bash
# mid-size batches (SURVEY section 8(d) C5 sweeps 8 / 16 / 32 windows per GPU): bench lines and the 32-window kernel profile
set -o pipefail
out=gpurun_out/r03w
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for b in 8 16 32 64; do
  python3 bench.py --batch $b --steps 3 --warmup 1 --no-cpu-baseline --no-overlap > $out/bench_b$b.json.log 2>&1; grep -E '^\{' $out/bench_b$b.json.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch', $b, d['value'], d['ms_per_step'], d['last_step_split_ms'])"
done
rocprofv3 --kernel-trace --output-format csv -d $out/prof -o t -- python3 bench.py --batch 32 --steps 1 --warmup 1 --no-cpu-baseline --no-overlap > $out/bench_b32_prof.json.log 2>&1 || exit 1
python3 scripts/trace_summary.py $(find $out/prof -name '*kernel_trace.csv' | head -1) > $out/trace_summary_b32.txt; rm -rf $out/prof
head -22 $out/trace_summary_b32.txt | cut -c1-125
