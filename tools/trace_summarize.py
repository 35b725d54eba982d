"""Prints the last diffusion step of a rocprofv3 kernel trace CSV: one line per launch (kernel, grid, duration)."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
last = max(i for i, n in enumerate(names) if "embed" in n)
tot = 0
for r in rows[last:]:
    n = r["Kernel_Name"]
    if "ed::" not in n:
        continue
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += dur
    short = n.split("::")[-1].split("(")[0][:48]
    print(f"{short:50s} grid={int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1):6d}x{int(r['Grid_Size_Y']) // max(int(r['Workgroup_Size_Y']), 1):4d} {dur:8.1f} us")
print("total", tot)
