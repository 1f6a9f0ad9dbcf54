#!/bin/bash
# PMC evidence for the self-attention's L1 stall: texture-address / L1 counters with plain and non-temporal K/V row loads (kernel test)
set -o pipefail
out=gpurun_out/r03t
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for nt in 0 1; do
  WRS_SELF_NT=$nt rocprofv3 --pmc TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $out/pmc_nt$nt -o p -- python3 -m pytest tests/test_gpu_attention.py -q -k "self_attention" > $out/pmc_nt$nt.log 2>&1 || { tail -20 $out/pmc_nt$nt.log; exit 1; }
  python3 scripts/pmc_summary.py $(find $out/pmc_nt$nt -name '*counter_collection.csv' | head -1) > $out/pmc_nt$nt.txt
  rm -rf $out/pmc_nt$nt
  echo "== WRS_SELF_NT=$nt"; grep -A4 "attn_self_mfma" $out/pmc_nt$nt.txt | head -12
done
