#!/bin/bash
# one PMC pass over the GEMM micro-benchmark: tools/pmc_quick.sh "<counters>" <gemm_bench args...>
set -u
ctrs=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmcq
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/gemm_bench.py --iters 3 "$@" > $out/log.txt 2>&1 || { echo "pass failed"; tail -5 $out/log.txt; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = []
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(agg.items())), f" dur_us={sum(dur)/max(len(dur),1):.1f}")
PY
