#!/bin/bash
# Memory-side traffic per launch of the weight-streaming kernels (ar_gemv.hip): FETCH_SIZE and WRITE_SIZE in separate
# --pmc passes over tools/ar_step.py on an 8-block model of the 640M widths.  usage: tools/pmc_ar_gemv.sh <B>
set -u
B=${1:-1}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_ar_b$B
mkdir -p $out
cd /tmp && export TMPDIR=/tmp ED_AR_LAYERS=8 ED_AR_BATCHES=$B
i=0
for p in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 170 rocprofv3 --pmc $p --output-format csv -d $out/t$i -- python3 $GRAFT_REPO_ROOT/tools/ar_step.py > $out/t$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/t$i.log; exit 1; }
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/t*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ar_gemv" not in k: continue
        agg[k[:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"  {c:14s} mean/dispatch = {sum(v)/len(v):.6g}  (n={len(v)})")
PY
rm -rf $out/t1 $out/t2   # (the per-dispatch csv files are large; the summary above is what is kept)
