#!/bin/bash
# 128-row tiles (kernel 2: library's tuned choice) against forced 256-row tiles (kernel 6) on mid-size problems
set -e
g() { timeout -k 10 60 python tools/gemm_bench.py "$@" --iters 100 --check 1 2>&1 | grep -E "us/launch|rel err" | paste - -; }
for B in 8 16 32; do
  for k in 2 6; do
    g --B $B --L 256 --flat 1 --taps 1 --resid 1 --kernel $k
    g --B $B --L 256 --flat 1 --taps 1 --out_bf16 1 --kernel $k
    for d in 1 16 32 64 128; do g --B $B --L 256 --taps 5 --dil $d --out_bf16 1 --kernel $k; done
  done
done
# ragged / multi-tile shapes through the 128-row forms
g --B 3 --L 200 --taps 5 --dil 64 --kernel 2 --iters 3
g --B 3 --L 77 --taps 5 --dil 32 --kernel 2 --iters 3
g --B 5 --L 256 --taps 5 --dil 128 --kernel 2 --iters 3
g --B 4 --L 512 --taps 5 --dil 32 --kernel 2 --iters 3
g --B 4 --L 512 --taps 5 --dil 64 --kernel 2 --iters 3
g --B 2 --L 1000 --N 512 --K 512 --taps 5 --dil 8 --kernel 2 --iters 3
g --B 2 --L 1000 --N 1024 --K 512 --taps 1 --flat 1 --resid 1 --kernel 2 --iters 3
g --B 16 --L 256 --N 512 --K 512 --taps 5 --dil 64 --kernel 2 --out_bf16 1 --iters 30
