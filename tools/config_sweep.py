"""OA-DM step time of BASELINE.json's other configurations (bf16, graph replay), with the per-kind kernel shares of
probed steps: config 1 (38M, B=1, L=100: the reference notebook's case), config 4 geometry (640M, L=512, B per sub-batch) and config 5 (38M, B=1024, L=1024).
    python tools/config_sweep.py [cfg1|cfg2|cfg4|cfg4b|cfg5] ..."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from evodiff_amd import pretrained  # noqa: E402

CASES = {"cfg1": ("38M", 1, 100, 100), "cfg2": ("640M", 64, 256, 32), "cfg4": ("640M", 64, 512, 32), "cfg4b": ("640M", 128, 512, 16),
         "cfg5": ("38M", 1024, 1024, 12)}
dev = torch.device("cuda", 0)
for name in (sys.argv[1:] or ["cfg4", "cfg5"]):
    size, B, L, n = CASES[name]
    torch.manual_seed(0)
    m, _, tok, _ = (pretrained.OA_DM_640M if size == "640M" else pretrained.OA_DM_38M)(random_init=True)
    m.precision = "bf16"
    order = torch.from_numpy(np.random.default_rng(0).permutation(L)[:n].astype(np.int32))
    uni = torch.rand((n, B), device=dev)
    best = 1e9
    for rep in range(3):
        sample = torch.full((B, L), tok.mask_id, dtype=torch.long, device=dev)
        if rep == 2:
            m.probe(4)
            m.probe_read()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.oadm_sample(sample, order, uni, 21)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        if rep < 2:
            best = min(best, dt)
    stats = m.probe_read()
    m.probe(0)
    hp = m.hparams
    D, H = hp["d_model"], hp["d_hidden"]
    f = 0.0
    for i in range(hp["n_layers"]):
        d = 2 ** (i % 8)
        f += L * 2.0 * 2 * D * H + 2.0 * H * H * sum(max(0, L - abs(o) * d) for o in range(-2, 3))
    f = (f + L * 2.0 * (8 * D + 31 * D)) * B
    print(f"{name}: {size} B={B} L={L}: {best * 1e3:.3f} ms/step, {f / best / 1e15:.3f} PFLOP/s of the full step, "
          f"{B / (best * L):.3f} seq/s", flush=True)
    print("   probed:", {k: (round(v["ms"], 3), v["launches"]) for k, v in stats.items()}, flush=True)
    m.release()
    del m
    torch.cuda.empty_cache()
