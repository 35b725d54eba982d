"""Autoregressive (LR_AR_640M, causal) sampling step time: one new row per layer and step from cached convolution
inputs (ed_ar_sample), against the reference's scheme of re-running the growing prefix (one ed_forward of length s+1
per step, timed at a few lengths)."""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from evodiff_amd.pretrained import LR_AR_640M  # noqa: E402

import os  # noqa: E402

torch.manual_seed(0)
if "ED_AR_LAYERS" in os.environ:   # a 640M-wide causal model with fewer blocks: 8 blocks = 184 MB of bf16 weights, which
    from evodiff_amd.model import ByteNetLMTime  # noqa: E402    stay in the 256 MB Infinity Cache between steps
    m = ByteNetLMTime(30, 8, 1280, int(os.environ["ED_AR_LAYERS"]), 5, 128, causal=True, padding_idx=27, final_ln=True,
                      slim=False, activation="gelu").eval()
else:
    m, _, tok, _ = LR_AR_640M(random_init=True)
m.precision = "bf16"
for opt, env in (("tile64_max", "ED_TILE64"), ("split_target", "ED_TARGET"), ("ar_gemv", "ED_AR_GEMV")):   # kernel-choice options, for A/B runs
    if env in os.environ:
        m.set_option(opt, int(os.environ[env]))
dev = torch.device("cuda", 0)
n = 256
wbytes = 2.0 * sum(p.numel() for p in m.parameters())
for B in [int(b) for b in os.environ.get("ED_AR_BATCHES", "1,4,8,64,256").split(",")]:
    uni = torch.rand((n, B), device=dev)
    best = 1e9
    for rep in range(3):
        sample = torch.full((B, n + 1), (29 if "ED_AR_LAYERS" in os.environ else tok.start_id), dtype=torch.long, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.ar_sample(sample, uni, n_steps=n)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n)
    print(f"ed_ar_sample B={B:4d}: {best * 1e3:.3f} ms/step, {B / best:.0f} tokens/s, weights/time = "
          f"{wbytes / best / 1e12:.2f} TB/s", flush=True)
for L in (() if "ED_AR_BATCHES" in os.environ else (64, 256, 1024)):
    x = torch.randint(0, 26, (1, L), device=dev)
    t = torch.zeros(1, dtype=torch.long, device=dev)
    m(x, t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        m(x, t)
    torch.cuda.synchronize()
    print(f"prefix re-run, B=1 L={L}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per forward", flush=True)
