"""Calibration only: what plain streaming kernels reach on the LayerNorm kernel's traffic shapes."""
import torch
dev = torch.device("cuda", 0)
def t(fn, n=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rows in (16384, 8192):
    xb = torch.randn((rows, 1280), device=dev).to(torch.bfloat16)
    xf = torch.randn((rows, 1280), device=dev)
    ob = torch.empty_like(xb)
    us = t(lambda: torch.mul(xb, 2, out=ob))
    print(f"rows={rows} bf16->bf16 scale: {us:.1f} us, {rows*1280*4/us/1e6:.2f} TB/s")
    us = t(lambda: ob.copy_(xf))
    print(f"rows={rows} fp32->bf16 convert: {us:.1f} us, {rows*1280*6/us/1e6:.2f} TB/s")
    us = t(lambda: ob.copy_(xb))
    print(f"rows={rows} bf16 copy: {us:.1f} us, {rows*1280*4/us/1e6:.2f} TB/s")
    of = torch.empty_like(xf)
    us = t(lambda: of.copy_(xf))
    print(f"rows={rows} fp32 copy: {us:.1f} us, {rows*1280*8/us/1e6:.2f} TB/s")
