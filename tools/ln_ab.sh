#!/bin/bash
for ib in 0 1; do
  timeout -k 10 60 python tools/ln_bench.py --in_bf16 $ib --out_bf16 1 --act 1 --iters 200 --check 1 2>&1 | grep -E "rows=|max abs" | paste - - || exit 1
done
for C in 512 640 1024 96 2048; do
timeout -k 10 60 python tools/ln_bench.py --in_bf16 1 --out_bf16 1 --act 1 --C $C --rows 5000 --iters 50 --check 1 2>&1 | grep -E "rows=|max abs" | paste - -
timeout -k 10 60 python tools/ln_bench.py --in_bf16 0 --out_bf16 1 --act 1 --C $C --rows 5000 --iters 50 --check 1 2>&1 | grep -E "rows=|max abs" | paste - -
done
