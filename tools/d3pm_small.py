"""D3PM step time and per-kernel-class split at small sizes (38M / 640M, B=1, L=100 / 256, bf16)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from evodiff_amd.pretrained import D3PM_UNIFORM_38M, D3PM_BLOSUM_640M  # noqa: E402

dev = torch.device("cuda", 0)
for name, loader, B, L in (("38M", D3PM_UNIFORM_38M, 1, 100), ("640M", D3PM_BLOSUM_640M, 1, 256), ("640M", D3PM_BLOSUM_640M, 64, 256)):
    torch.manual_seed(0)
    m, _, tok, _, T, Qbar, Q = loader(return_all=True, random_init=True)
    m.precision = "bf16"
    K, n = tok.K, 32
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
    m.probe(1)
    sample = start.clone()
    m.d3pm_sample(sample, Qd, Qbd, uni, K, t_start=T - 1, n_steps=4)
    st = m.probe_read()
    m.probe(0)
    print(f"D3PM {name} B={B} L={L}: {best * 1e3:.3f} ms/step;  probed ms/step:", {k: round(v["ms"] / 4, 3) for k, v in st.items()}, flush=True)
    m.release()
