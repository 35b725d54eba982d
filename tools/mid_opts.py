"""Mid-batch OA-DM step time (640M, L=256, bf16, graph replay) under the kernel-choice options
ED_OPT_MIN_TUNED_TILES x ED_OPT_TILE64_MAX (x ED_OPT_PARTIALS_MB)."""
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
L, n = 256, 32
batches = [int(b) for b in sys.argv[1].split(",")] if len(sys.argv) > 1 else [16, 24, 32, 48]
combos = [(130, 320, 32), (64, 320, 32), (100, 320, 32), (200, 320, 32), (130, 160, 32), (130, 640, 32)]
for B in batches:
    order = torch.from_numpy(np.random.default_rng(0).permutation(L)[:n].astype(np.int32))
    uni = torch.rand((n, B), device=dev)
    for mtt, t64, pmb in combos:
        m.set_option("partials_mb", pmb)
        m.set_option("tile64_max", t64)
        m.set_option("min_tuned_tiles", mtt)
        best = 1e9
        for rep in range(3):
            sample = torch.full((B, L), tok.mask_id, dtype=torch.long, device=dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m.oadm_sample(sample, order, uni, 21)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / n)
        print(f"B={B:3d} min_tuned_tiles={mtt} tile64_max={t64} partials_mb={pmb}: {best * 1e3:.3f} ms/step", flush=True)
