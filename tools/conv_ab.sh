#!/bin/bash
# windowed convolution (kernel 2; 9 = with in-kernel clock stamps) against the per-tap form (kernel 5)
for d in 1 4 16; do
  for k in 9 5; do
    timeout -k 10 60 python tools/gemm_bench.py --B 64 --L 256 --taps 5 --dil $d --kernel $k --out_bf16 1 --iters 100 --check 1 2>&1 | grep -E "us/launch|rel err|K loop" || exit 1
  done
done
timeout -k 10 60 python tools/gemm_bench.py --B 3 --L 200 --taps 5 --dil 2 --kernel 2 --out_bf16 0 --iters 3 --check 1 2>&1 | grep -E "us/launch|rel err" || exit 1
timeout -k 10 60 python tools/gemm_bench.py --B 64 --L 256 --N 512 --K 512 --taps 5 --dil 4 --kernel 2 --out_bf16 1 --iters 30 --check 1 2>&1 | grep -E "us/launch|rel err" || exit 1
timeout -k 10 60 python tools/gemm_bench.py --B 64 --L 256 --N 512 --K 512 --taps 5 --dil 4 --kernel 5 --out_bf16 1 --iters 30 --check 1 2>&1 | grep -E "us/launch|rel err" || exit 1
