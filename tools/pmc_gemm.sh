#!/bin/bash
# PMC passes over the implicit-GEMM micro-benchmark (one pass per run; counters per block within limits).
# usage: tools/pmc_gemm.sh <tag> <gemm_bench args...>
set -u
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
p1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT"
p2="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"
p3="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"
i=0
for p in "$p1" "$p2" "$p3"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $p --output-format csv -d $out/p$i -- python3 $GRAFT_REPO_ROOT/tools/gemm_bench.py --iters 3 "$@" > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; exit 1; }
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm" not in k: continue
        agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"  {c:34s} mean/dispatch = {sum(v)/len(v):.4g}  (n={len(v)})")
PY
