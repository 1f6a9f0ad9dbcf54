#!/bin/bash
# MFMA-utilisation counters for the encoder GEMMs / attention and the decode GEMMs (north star: "MFMA-utilisation counters against gfx950 peak")
set -o pipefail
out=gpurun_out/r03y
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 -L > $out/counters_list.txt 2>&1; grep -i -E "MFMA|GRBM_GUI_ACTIVE|SQ_BUSY_CYCLES|SQ_BUSY_CU" $out/counters_list.txt | head -30
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_mfma -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap --no-shard-check --sample-len 24 > $out/pmc_mfma.log 2>&1 || { tail -5 $out/pmc_mfma.log; exit 1; }
python3 scripts/pmc_summary.py $(find $out/pmc_mfma -name '*counter_collection.csv' | head -1) > $out/pmc_mfma.txt; rm -rf $out/pmc_mfma
grep -A4 -E "gemm_f16|attn_enc|attn_mfma|attn_self_mfma" $out/pmc_mfma.txt | head -80
