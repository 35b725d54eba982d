"""Per-diffusion-step summary of a rocprofv3 kernel_stats.csv of `bench.py --no-graph --steps S --warmup W`:
calls and microseconds per step by kernel (template arguments kept, argument lists dropped).
    python tools/stats_per_step.py <kernel_stats.csv> <number of diffusion steps in the run>"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
total = 0.0
out = []
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::|ed::|void ", "", r["Name"])
    name = re.sub(r"\(.*$", "", name)[:70]
    us = float(r["TotalDurationNs"]) / 1e3 / steps
    total += us
    out.append((us, name, float(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3))
for us, name, calls, avg in sorted(out, reverse=True)[:24]:
    print(f"{us:9.1f} us/step  {calls:7.2f} calls/step  avg {avg:7.1f} us  {name}")
print(f"{total:9.1f} us/step in kernels")
