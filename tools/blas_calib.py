"""Calibration only (not part of the product path): what the vendor GEMM library reaches on the block's shapes."""
import torch

dev = torch.device("cuda", 0)
for (M, N, K) in ((16384, 1280, 1280), (16384, 1280, 6400), (8192, 1280, 6400), (4096, 1280, 6400), (16384, 512, 2560)):
    a = (torch.rand((M, K), device=dev) * 2 - 1).to(torch.bfloat16)
    w = (torch.rand((N, K), device=dev) * 2 - 1).to(torch.bfloat16)
    for _ in range(5):
        c = a @ w.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        c = a @ w.t()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    print(f"M={M} N={N} K={K}: {ms * 1e3:.1f} us, {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s", flush=True)
