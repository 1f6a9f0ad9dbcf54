#!/bin/bash
# round-3 batch 8: smoke() on the two-library build, config tests, filter tests
out=gpurun_out/r03i
mkdir -p $out
python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $out/smoke.log | cut -c1-300
timeout -k 10 600 python3 -m pytest tests/test_gpu_configs.py tests/test_gpu_filters.py tests/test_gpu_example.py -q > $out/pytest_cfg.log 2>&1; tail -5 $out/pytest_cfg.log
