"""D3PM_BLOSUM_640M step time (BASELINE config 3: B=64, L=256, bf16) and its per-kernel-class split."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from evodiff_amd.pretrained import D3PM_BLOSUM_640M  # noqa: E402

torch.manual_seed(0)
m, _, tok, _, T, Qbar, Q = D3PM_BLOSUM_640M(return_all=True, random_init=True)
m.precision = "bf16"
dev = torch.device("cuda", 0)
B, L, K, n = 64, 256, tok.K, 16
start = torch.from_numpy(np.random.default_rng(2).integers(0, K, (B, L))).to(dev)
uni = torch.rand((T, B, L), device=dev)
Qd, Qbd = Q.to(dev), Qbar.to(dev)
best = 1e9
for rep in range(3):
    sample = start.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.d3pm_sample(sample, Qd, Qbd, uni, K, t_start=T - 1, n_steps=n)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / n)
print(f"D3PM 640M B={B} L={L}: {best * 1e3:.3f} ms/step -> {B / (best * (T - 1)):.3f} seq/s (499 steps per sequence)")
m.probe(1)
sample = start.clone()
m.d3pm_sample(sample, Qd, Qbd, uni, K, t_start=T - 1, n_steps=4)
st = m.probe_read()
print({k: round(v["ms"] / 4, 3) for k, v in st.items()})
