"""The calls of the reference's examples/evodiff.ipynb (cells 8-14) through the drop-in API, timed: OA_DM_38M
generate_oaardm and D3PM_UNIFORM_38M generate_d3pm, batch 1, length 100 (stored notebook outputs: 10 s and 57 s on a
laptop CPU).  Random-init weights (no checkpoints offline)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from evodiff_amd.pretrained import OA_DM_38M, D3PM_UNIFORM_38M  # noqa: E402
from evodiff_amd.generate import generate_oaardm, generate_d3pm  # noqa: E402

torch.manual_seed(0)
np.random.seed(0)
model, collater, tokenizer, scheme = OA_DM_38M(random_init=True)
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tokens, strings = generate_oaardm(model, tokenizer, 100, batch_size=1, device="cuda")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"OA_DM_38M generate_oaardm B=1 L=100: {dt:.3f} s ({100 / dt:.0f} it/s)  {strings[0][:40]}...")

model, collater, tokenizer, scheme, timestep, Q_prod, Q_t = D3PM_UNIFORM_38M(return_all=True, random_init=True)
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tokens, strings = generate_d3pm(model, tokenizer, Q_t, Q_prod, timestep, 100, batch_size=1, device="cuda")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"D3PM_UNIFORM_38M generate_d3pm B=1 L=100: {dt:.3f} s ({499 / dt:.0f} it/s)  {strings[0][:40]}...")
