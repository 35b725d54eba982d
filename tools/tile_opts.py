"""OA-DM step time (640M, L=256, bf16, graph replay) under ED_OPT_TILE64_MAX x ED_OPT_SPLIT_TARGET."""
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
batches = [int(b) for b in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 4, 8, 64]
combos = [tuple(int(v) for v in c.split(":")) for c in sys.argv[2].split(",")] if len(sys.argv) > 2 else [(0, 512), (40, 512), (160, 1024), (320, 1024)]
for B in batches:
    n = 48 if B < 32 else 32
    order = torch.from_numpy(np.random.default_rng(0).permutation(L)[:n].astype(np.int32))
    uni = torch.rand((n, B), device=dev)
    for t64, target in combos:
        m.set_option("tile64_max", t64)
        m.set_option("split_target", target)
        best = 1e9
        for rep in range(3):
            sample = torch.full((B, L), tok.mask_id, dtype=torch.long, device=dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m.oadm_sample(sample, order, uni, 21)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / n)
        print(f"B={B:3d} tile64_max={t64} target={target}: {best * 1e3:.3f} ms/step", flush=True)
