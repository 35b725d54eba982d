#!/bin/bash
# One GPU-box visit: correctness of the small-tile kernel on ragged shapes, the GPU parity suite, the small- and
# mid-batch timing tools, the bench line and its rocprofv3 kernel stats.  Every step has its own time limit and the
# script stops at the first failure.  Logs under gpurun_out/.
set -u
cd "$GRAFT_REPO_ROOT"
out=gpurun_out
step() { echo "== $1"; shift; "$@" || { echo "FAILED: $*"; exit 1; }; }
step "gemm checks" bash -c "timeout -k 10 60 python tools/gemm_bench.py --B 1 --L 256 --taps 1 --kernel 105 --out_bf16 1 --iters 20 --check 1 && timeout -k 10 60 python tools/gemm_bench.py --B 1 --L 256 --taps 5 --kernel 116 --out_bf16 1 --iters 20 --check 1 && bash tools/small_rows.sh" 2>&1 | grep -v amdgpu.ids | tee $out/s_gemm.log
grep -q FAILED $out/s_gemm.log && exit 1
step "pytest gpu" bash -c "timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $out/s_pytest.log 2>&1; rc=\$?; tail -4 $out/s_pytest.log; exit \$rc"
step "small batch" bash -c "timeout -k 10 200 python -u tools/small_batch.py 2>&1 | grep -v amdgpu.ids | tee $out/s_small.log"
step "configs" bash -c "timeout -k 10 300 python -u tools/config_sweep.py cfg1 cfg2 2>&1 | grep -v amdgpu.ids | tee $out/s_cfg.log"
step "ar" bash -c "timeout -k 10 300 python -u tools/ar_step.py 2>&1 | grep -v amdgpu.ids | tee $out/s_ar.log"

step "bench" bash -c "timeout -k 10 300 python bench.py > $out/s_bench.json 2> $out/s_bench.err; rc=\$?; cat $out/s_bench.json; exit \$rc"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
step "rocprof stats" bash -c "timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/s_prof -o bench -- python3 bench.py --no-graph --no-cpu-baseline --steps 1 --warmup 1 > $out/s_prof.log 2>&1; rc=\$?; tail -2 $out/s_prof.log; ls $out/s_prof; exit \$rc"
echo "== done"
