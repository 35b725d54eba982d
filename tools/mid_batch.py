import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from evodiff_amd.pretrained import OA_DM_640M
torch.manual_seed(0)
m, _, tok, _ = OA_DM_640M(random_init=True)
m.precision = "bf16"
dev = torch.device("cuda", 0)
L, n = 256, 24
for B in (16, 24, 32, 48, 64):
    for tiles in (70, 130, 140):
        m.set_option("min_tuned_tiles", tiles)
        order = torch.from_numpy(np.random.default_rng(0).permutation(L)[:n].astype(np.int32))
        uni = torch.rand((n, B), device=dev)
        best = 1e9
        for rep in range(3):
            sample = torch.full((B, L), tok.mask_id, dtype=torch.long, device=dev)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m.oadm_sample(sample, order, uni, 21)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / n)
        print(f"B={B:3d} min_tuned_tiles={tiles:4d}: {best*1e3:.3f} ms/step {B/(best*L):.3f} seq/s", flush=True)
