#!/bin/bash
# HBM-side traffic of ONE kernel launch: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (TCC counter budget),
# over the implicit-GEMM micro-benchmark.  usage: tools/pmc_traffic.sh <tag> <gemm_bench args...>
set -u
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for p in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $p --output-format csv -d $out/t$i -- python3 $GRAFT_REPO_ROOT/tools/gemm_bench.py --iters 3 "$@" > $out/t$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/t$i.log; exit 1; }
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/t*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ed::" not in k: continue
        agg[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"  {c:14s} mean/dispatch = {sum(v)/len(v):.6g}  (n={len(v)})")
PY
