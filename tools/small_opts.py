"""Small-batch OA-DM step time (640M, L=256, bf16, graph replay) under the split-K options:
ED_OPT_FUSE_REDUCE_LN x ED_OPT_SPLIT_MIN_ITERS x ED_OPT_SPLIT_TARGET.  Also the autoregressive step (LR_AR_640M)."""
import itertools
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
L, n = 256, 48
batches = [int(b) for b in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 4, 16]
combos = [(0, 4, 512), (1, 4, 512), (1, 2, 512), (1, 1, 512), (1, 2, 1024), (1, 1, 1024)]
for B in batches:
    order = torch.from_numpy(np.random.default_rng(0).permutation(L)[:n].astype(np.int32))
    uni = torch.rand((n, B), device=dev)
    for fuse, min_iters, target in combos:
        m.set_option("fuse_reduce_ln", fuse)
        m.set_option("split_min_iters", min_iters)
        m.set_option("split_target", target)
        best = 1e9
        for rep in range(3):
            sample = torch.full((B, L), tok.mask_id, dtype=torch.long, device=dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m.oadm_sample(sample, order, uni, 21)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / n)
        print(f"B={B:3d} fuse={fuse} min_iters={min_iters} target={target}: {best * 1e3:.3f} ms/step", flush=True)
