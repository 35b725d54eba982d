"""Times OA-DM steps (640M, B=64, L=256, bf16, graph replay) under the execution options of the C ABI."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from evodiff_amd.pretrained import OA_DM_640M  # noqa: E402

torch.manual_seed(0)
m, _, tok, _ = OA_DM_640M(random_init=True)
m.precision = "bf16"
dev = torch.device("cuda", 0)
B, L, n = 64, 256, 32
order = torch.from_numpy(np.random.default_rng(0).permutation(L)[:n].astype(np.int32))
uni = torch.rand((n, B), device=dev)
for prune, tiles, split in ((1, 40, 512), (1, 24, 512), (1, 64, 512), (1, 40, 128), (1, 40, 2048), (0, 40, 512)):
    m.set_option("prune", prune)
    m.set_option("min_tuned_tiles", tiles)
    m.set_option("split_target", split)
    best = 1e9
    for rep in range(3):
        sample = torch.full((B, L), tok.mask_id, dtype=torch.long, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.oadm_sample(sample, order, uni, 21)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n)
    print(f"prune={prune} min_tuned_tiles={tiles:4d} split_target={split:5d}: {best * 1e3:.3f} ms/step -> "
          f"{B / (best * L):.3f} seq/s", flush=True)
