#!/bin/bash
# Matrix-pipe utilisation of the large kernels at config-2 size (B=64, L=256, 640M widths) from SQ counters: one
# --pmc pass per kernel over the implicit-GEMM micro-benchmark.  usage: tools/pmc_mfma.sh   (writes gpurun_out/pmc_mfma.txt)
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_mfma
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
ctrs="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
run() {
  tag=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/$tag -- python3 $GRAFT_REPO_ROOT/tools/gemm_bench.py --iters 20 "$@" > $out/$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 $out/$tag.log; exit 1; }
}
run conv_d1   --B 64 --L 256 --taps 5 --dil 1   --kernel ${PMC_KERNEL:-0} --out_bf16 1
run conv_d16  --B 64 --L 256 --taps 5 --dil 16  --kernel ${PMC_KERNEL:-0} --out_bf16 1
run conv_d64  --B 64 --L 256 --taps 5 --dil 64  --kernel ${PMC_KERNEL:-0} --out_bf16 1
run pff1      --B 64 --L 256 --flat 1 --taps 1 --kernel ${PMC_KERNEL:-0} --out_bf16 1
run pff2      --B 64 --L 256 --flat 1 --taps 1 --resid 1 --kernel ${PMC_KERNEL:-0}
python3 - "$out" <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/pmc_mfma.txt
import csv, glob, os, sys, collections
out = sys.argv[1]
CUS, SIMDS, XCDS = 256, 4, 8
print("tag        kernel                                   dur_us  clk_GHz  MFMA_insts  MFMA_busy_cyc/SIMD  kernel_cyc  mfma_busy_frac  lds_conflict/lds_active")
for tag in ("conv_d1", "conv_d16", "conv_d64", "pff1", "pff2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "ed::" in r["Kernel_Name"]:
                agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{tag}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "ed::" in r["Kernel_Name"]:
                dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, d in agg.items():
        m = {c: sum(v) / len(v) for c, v in d.items()}
        us = sum(dur[k]) / max(len(dur[k]), 1)
        kcyc = m.get("GRBM_GUI_ACTIVE", 0) / XCDS
        busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (CUS * SIMDS)
        name = k.split("::")[-1][:40]
        print(f"{tag:10s} {name:40s} {us:7.1f} {kcyc / max(us, 1e-9) / 1e3:7.2f} {m.get('SQ_INSTS_MFMA', 0):11.4g} "
              f"{busy:19.4g} {kcyc:11.4g} {busy / max(kcyc, 1):15.3f} "
              f"{m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m.get('SQ_LDS_IDX_ACTIVE', 1), 1):10.4f}")
PY
