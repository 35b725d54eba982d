"""Random-geometry parity sweep on the GPU (a tool: the suite holds six of its cases as regression).  Each seed draws a
model geometry (width, slim or not, depth, dilation schedule, GELU / ReLU, causal or not, with or without timesteps) and a
batch / length and runs tests/test_gpu_parity.py::random_geometry_case on it (forward fp32 / bf16, OA-DM loop with and
without pruning / incremental layers / graph replay, cached autoregressive loop, all against the CPU oracle).
Usage: python tools/fuzz_parity.py [first_seed [count]]; prints one line per case and exits non-zero on the first failure."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_parity import random_geometry_case  # noqa: E402

dev = torch.device("cuda", 0)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 24
t0 = time.time()
for seed in range(first, first + count):
    print(random_geometry_case(seed, dev), flush=True)
print(f"{count} cases in {time.time() - t0:.0f} s")
