#!/usr/bin/env python3
"""Experiment: does sampling two half-batches concurrently on two HIP streams beat one full batch?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evodiff_amd import pretrained

dev = torch.device("cuda", 0)
torch.manual_seed(0)
nsplit = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B, L = 64, 256
models = []
for i in range(nsplit):
    torch.manual_seed(0)
    m, _, tok, _ = pretrained.OA_DM_640M(random_init=True)
    m.precision = "bf16"
    models.append(m)
streams = [torch.cuda.Stream() for _ in range(nsplit)]
Bs = B // nsplit
rng = np.random.default_rng(0)
order = torch.from_numpy(rng.permutation(L).astype(np.int32)).to(dev)
uni = torch.rand((L, Bs), device=dev)
def run():
    samples = [torch.full((Bs, L), tok.mask_id, dtype=torch.long, device=dev) for _ in range(nsplit)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for m, s, x in zip(models, streams, samples):
        with torch.cuda.stream(s):
            m.oadm_sample(x, order, uni, 21)
    torch.cuda.synchronize()
    return time.perf_counter() - t0
run()
for _ in range(2):
    dt = run()
    print(f"nsplit={nsplit}: {dt:.3f} s per {B} sequences -> {B/dt:.2f} seq/s")
