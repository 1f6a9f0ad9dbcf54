#!/bin/bash
# round-3 final: default bench FIRST (cool GPU, as the driver's fresh box), round profile (sequential trace + PMC passes), full GPU suite, smoke
set -o pipefail
out=gpurun_out/r03final
mkdir -p $out
bash scripts/profile_round.sh r03final > $out/profile.log 2>&1 || { tail -5 $out/profile.log; exit 1; }
grep -E '^\{' $out/bench_default.json.log | cut -c1-700
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"
