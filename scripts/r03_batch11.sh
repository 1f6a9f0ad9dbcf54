#!/bin/bash
# does the cross-attention's rate depend on the allocation or on the device's state?  4 processes x 5 fresh allocations, clocks read in between
out=gpurun_out/r03m
mkdir -p $out
for p in 1 2 3; do
  echo "== process $p" >> $out/cross_state.log
  (rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "mclk|fclk|sclk|socclk|Power|Temperature" | head -12) >> $out/cross_state.log 2>&1
  timeout -k 10 120 scripts/micro/cross_attn_state 307 5 >> $out/cross_state.log 2>&1 || exit 1
done
cat $out/cross_state.log
