"""generate_oaardm_order_opt, OA_DM_38M, L = 100, B = 1 (generate.py:151-177): one ed_forward per step plus the
argmax / draw as device ops.  Run under `rocprofv3 --hip-trace --stats` to count host synchronisations per step."""
import sys
import time

import torch

sys.path.insert(0, ".")
from evodiff_amd import pretrained  # noqa: E402
from evodiff_amd.generate import generate_oaardm_order_opt  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
m, _, tok, _ = pretrained.OA_DM_38M(random_init=True)
m.precision = "bf16"
generate_oaardm_order_opt(m, tok, 100, batch_size=1, device=dev)       # warm-up: weight upload, workspace
torch.cuda.synchronize()
print("MARK: timed loop starts", flush=True)
t0 = time.perf_counter()
sample, strings = generate_oaardm_order_opt(m, tok, 100, batch_size=1, device=dev)
dt = time.perf_counter() - t0
m.check_errors()
print(f"order_opt 38M L=100 B=1: {dt * 1e3:.1f} ms for 100 steps; {strings[0][:40]}...")
