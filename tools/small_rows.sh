#!/bin/bash
# conv and 1x1 GEMM at shrinking row counts, tuned (2) vs generic (1) kernel
for rows in 64 32 16 8 4 2 1; do
  for k in 1 2; do
    timeout -k 10 60 python tools/gemm_bench.py --B $rows --L 256 --flat 1 --taps 5 --dil 1 --kernel $k --out_bf16 1 --iters 30 2>&1 | grep us/launch || exit 1
  done
done
for rows in 64 32 16 8 4 2 1; do
  for k in 1 2; do
    timeout -k 10 60 python tools/gemm_bench.py --B $rows --L 256 --flat 1 --taps 1 --kernel $k --out_bf16 1 --iters 30 2>&1 | grep us/launch || exit 1
  done
done
