#!/bin/bash
# Per-launch time of the tuned kernels with and without the LayerNorm in their epilogue, and with parts of that
# epilogue switched off (ed_debug_gemm kernel 400 + bits: 1 no wait, 2 no GELU, 4 no stores, 8 no statistics pass).
#   tools/ln_fuse_ab.sh [shapes: pff1 pff2 conv1 conv64]
declare -A SHAPES=( [pff1]="--taps 1 --flat 1 --out_bf16 1" [pff2]="--taps 1 --flat 1 --resid 1" [conv1]="--taps 5 --dil 1 --out_bf16 1" [conv64]="--taps 5 --dil 64 --out_bf16 1" )
for name in ${@:-pff1 pff2 conv1 conv64}; do
  for k in 2 400 401 402 404 408 415; do
    echo -n "$name "
    timeout -k 10 60 python tools/gemm_bench.py ${SHAPES[$name]} --kernel $k --iters 200 2>&1 | grep -v -e amdgpu.ids -e "^stamps" || exit 1
  done
done
