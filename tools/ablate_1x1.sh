#!/bin/bash
# What bounds the tuned kernels' K loops: the same launch with its MFMAs left out (copies + fragment reads only) and
# with its LDS-DMA copies left out (MFMAs on stale LDS only), next to the real one.  The ablated forms (kernels 3 / 4)
# store fp32 and take no residual; their results are invalid by construction.
set -e
g() { timeout -k 10 90 python tools/gemm_bench.py "$@" --iters 200 2>&1 | grep -E "us/launch|error"; }
echo "== 1x1 shape (16384 x 1280 x 1280, fp32 out): real / no MFMAs / no copies"
for k in 2 3 4; do g --B 64 --L 256 --flat 1 --taps 1 --kernel $k; done
echo "== 1x1 shape as the step runs it: bf16 out (PFF1), fp32 residual (PFF2)"
g --B 64 --L 256 --flat 1 --taps 1 --out_bf16 1 --kernel 2
g --B 64 --L 256 --flat 1 --taps 1 --resid 1 --kernel 2
echo "== convolution d = 1, per-tap form: real (kernel 5) / no MFMAs / no copies; windowed form (kernel 2)"
for k in 5 3 4; do g --B 64 --L 256 --taps 5 --dil 1 --kernel $k; done
g --B 64 --L 256 --taps 5 --dil 1 --out_bf16 1 --kernel 2
