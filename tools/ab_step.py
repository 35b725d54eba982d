"""Same-box A/B of two library builds: times OA-DM steps (640M, B=64, L=256, bf16) with $EVODIFF_AMD_LIB unset
and set (each in its own process, alternating), and prints ms per diffusion step."""
import os
import subprocess
import sys

CHILD = r'''
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from evodiff_amd.pretrained import OA_DM_640M
torch.manual_seed(0)
m, _, tok, _ = OA_DM_640M(random_init=True)
m.precision = "bf16"
dev = torch.device("cuda", 0)
B, L, n = 64, 256, 24
order = torch.from_numpy(np.random.default_rng(0).permutation(L)[:n].astype(np.int32))
uni = torch.rand((n, B), device=dev)
best = 1e9
for rep in range(4):
    sample = torch.full((B, L), tok.mask_id, dtype=torch.long, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.oadm_sample(sample, order, uni, 21)
    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / n)
print(f"{best * 1e3:.3f}")
'''
other = sys.argv[1]
for rnd in range(2):
    for name, lib in (("tree", None), ("other", other)):
        env = dict(os.environ)
        if lib:
            env["EVODIFF_AMD_LIB"] = os.path.abspath(lib)
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
        if out.returncode != 0:
            print(out.stderr[-2000:])
            sys.exit(1)
        print(f"{name:6s} {out.stdout.strip().splitlines()[-1]} ms/step", flush=True)
