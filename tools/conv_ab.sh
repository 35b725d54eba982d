#!/bin/bash
# windowed convolution (kernel 2) against the per-tap form (kernel 5), each checked against torch fp32
set -e
g() { timeout -k 10 60 python tools/gemm_bench.py "$@" --check 1 2>&1 | grep -E "us/launch|rel err" | paste - -; }
for d in 1 16 32 64 128; do
  for k in 2 5; do g --B 64 --L 256 --taps 5 --dil $d --kernel $k --out_bf16 1 --iters 100; done
done
g --B 3 --L 200 --taps 5 --dil 64 --kernel 2 --iters 3
g --B 3 --L 77 --taps 5 --dil 32 --kernel 2 --iters 3
g --B 5 --L 256 --taps 5 --dil 128 --kernel 2 --iters 3
g --B 64 --L 256 --N 512 --K 512 --taps 5 --dil 64 --kernel 2 --out_bf16 1 --iters 30
g --B 64 --L 256 --N 512 --K 512 --taps 5 --dil 64 --kernel 5 --out_bf16 1 --iters 30
g --B 4 --L 512 --taps 5 --dil 64 --kernel 2 --iters 3
g --B 2 --L 1024 --N 512 --K 512 --taps 5 --dil 8 --kernel 2 --iters 3
