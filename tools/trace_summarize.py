"""Prints the last diffusion step of a rocprofv3 kernel trace CSV: one line per launch (kernel, grid, duration)."""
import csv
import re
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
last = max(i for i, n in enumerate(names) if "embed" in n)
tot = 0
prev_end = None
for r in rows[last:]:
    n = r["Kernel_Name"]
    if "ed::" not in n:
        continue
    t0, t1 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur = (t1 - t0) / 1e3
    gap = (t0 - prev_end) / 1e3 if prev_end is not None else 0.0
    prev_end = t1
    tot += dur
    short = re.sub(r"\(anonymous namespace\)::|ed::|void ", "", n)
    short = re.sub(r"\(.*$", "", short)[:58]
    g = [int(r[f"Grid_Size_{a}"]) // max(int(r[f"Workgroup_Size_{a}"]), 1) for a in "XYZ"]
    print(f"{short:60s} grid={g[0]:5d}x{g[1]:4d}x{g[2]:2d} {dur:8.1f} us  (gap {gap:5.1f})")
print("total", round(tot, 1))
