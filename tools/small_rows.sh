#!/bin/bash
# generic (1) vs tuned (2) kernel at shrinking row counts; correctness of the generic kernel on ragged shapes
set -e
g() { timeout -k 10 60 python tools/gemm_bench.py "$@" --check 1 2>&1 | grep -E "us/launch|rel err" | paste - -; }
for rows in 16 8 4 2 1; do
  g --B $rows --L 256 --flat 1 --taps 5 --dil 1 --kernel 1 --out_bf16 1 --iters 30
done
for rows in 16 4 1; do
  g --B $rows --L 256 --flat 1 --taps 1 --kernel 1 --out_bf16 1 --iters 30
  g --B $rows --L 256 --flat 1 --taps 1 --kernel 1 --resid 1 --iters 30
done
g --B 3 --L 77 --N 96 --K 96 --taps 5 --dil 3 --kernel 1 --iters 3
g --B 2 --L 130 --N 200 --K 160 --taps 1 --kernel 1 --resid 1 --iters 3
