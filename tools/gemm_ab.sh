#!/bin/bash
set -e
g() { timeout -k 10 60 python tools/gemm_bench.py "$@" --iters 100 --check 1 2>&1 | grep -E "us/launch|rel err|stamps"; }
g --B 64 --L 256 --flat 1 --taps 1 --resid 1 --kernel 9
g --B 64 --L 256 --flat 1 --taps 1 --resid 1 --kernel 2
g --B 64 --L 256 --flat 1 --taps 1 --N 1024 --K 512 --resid 1 --kernel 2
