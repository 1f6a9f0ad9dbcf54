#!/bin/bash
# Round profile on the GPU box: default bench line, rocprofv3 kernel stats (sequential + overlapped passes), PMC passes
# (FETCH_SIZE / WRITE_SIZE in separate runs, as /opt/skills/guides/MI355X_MICROARCH.md prescribes).  Writes under
# gpurun_out/$1/; copy what should be judged into profiles/.
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash scripts/profile_round.sh r02'
set -o pipefail
tag=${1:-rXX}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
if [ -z "$SKIP_BENCH" ]; then python3 bench.py --steps 3 --warmup 1 > $out/bench_default.json.log 2>&1 && echo "bench default done" || exit 1; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_seq -o seq -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap > $out/bench_under_rocprof_sequential.json.log 2>&1 && echo "trace seq done" || exit 1
python3 scripts/trace_summary.py $(find $out/prof_seq -name '*kernel_trace.csv' | head -1) > $out/trace_summary_sequential.txt && echo "summary done"
cp $(find $out/prof_seq -name '*kernel_stats.csv' | head -1) $out/kernel_stats_sequential.csv
rm -rf $out/prof_seq
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap --sample-len 24 > $out/pmc_$c.log 2>&1 && echo "pmc $c done" || exit 1
  python3 scripts/pmc_summary.py $(find $out/pmc_$c -name '*counter_collection.csv' | head -1) > $out/pmc_$c.txt
  rm -rf $out/pmc_$c
done
ls -la $out
