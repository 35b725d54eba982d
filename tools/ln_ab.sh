#!/bin/bash
for wpb in 1 2 4 8 16; do
  for ib in 0 1; do
    EVODIFF_AMD_LN_WPB=$wpb timeout -k 10 60 python tools/ln_bench.py --in_bf16 $ib --out_bf16 1 --act 1 --iters 200 2>&1 | grep -E "rows=" | sed "s/^/wpb=$wpb /" || exit 1
  done
done
