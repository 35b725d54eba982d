"""OA-DM step time (640M, L=256, bf16, graph replay) under ED_OPT_TUNED_SPLIT_MIN_ROWS (x ED_OPT_PARTIALS_MB)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from evodiff_amd import pretrained  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
m, _, tok, _ = pretrained.OA_DM_640M(random_init=True)
m.precision = "bf16"
L = 256
batches = [int(b) for b in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2, 4, 8, 16, 24, 64]
combos = [(0, 32), (256, 128), (512, 128), (1024, 128), (2048, 128), (4096, 128)]
for B in batches:
    n = 32
    order = torch.from_numpy(np.random.default_rng(0).permutation(L)[:n].astype(np.int32))
    uni = torch.rand((n, B), device=dev)
    for rows, pmb in combos:
        m.set_option("tuned_split_min_rows", rows)
        m.set_option("partials_mb", pmb)
        best = 1e9
        for rep in range(3):
            sample = torch.full((B, L), tok.mask_id, dtype=torch.long, device=dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m.oadm_sample(sample, order, uni, 21)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / n)
        print(f"B={B:3d} tuned_split_min_rows={rows} partials_mb={pmb}: {best * 1e3:.3f} ms/step", flush=True)
