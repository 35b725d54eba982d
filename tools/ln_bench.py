#!/usr/bin/env python3
"""Micro-benchmark of the LayerNorm+activation kernel through the C ABI (ed_debug_ln_act)."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evodiff_amd import _lib

ap = argparse.ArgumentParser()
for n, d in (("rows", 16384), ("C", 1280), ("in_bf16", 0), ("out_bf16", 1), ("act", 1), ("iters", 50), ("check", 0)):
    ap.add_argument("--" + n, type=int, default=d)
a = ap.parse_args()
lib = _lib.load()
dev = torch.device("cuda", 0)
x = torch.randn((a.rows, a.C), device=dev) * 2 + 0.3
xin = x.to(torch.bfloat16) if a.in_bf16 else x
out = torch.empty((a.rows, a.C), dtype=torch.bfloat16 if a.out_bf16 else torch.float32, device=dev)
g = torch.rand(a.C, device=dev) + 0.5
b = torch.rand(a.C, device=dev) - 0.5
st = torch.cuda.current_stream().cuda_stream
def go():
    _lib.check(lib.ed_debug_ln_act(xin.data_ptr(), a.in_bf16, out.data_ptr(), a.out_bf16, g.data_ptr(), b.data_ptr(),
                                   a.rows, a.C, a.act, st))
for _ in range(5): go()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters): go()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.iters
nbytes = a.rows * a.C * ((2 if a.in_bf16 else 4) + (2 if a.out_bf16 else 4))
print(f"rows={a.rows} C={a.C} in_bf16={a.in_bf16} out_bf16={a.out_bf16} act={a.act}: {ms*1e3:.1f} us, {nbytes/ms/1e9:.2f} TB/s")
if a.check:
    ref = torch.nn.functional.layer_norm(xin.float(), (a.C,), g, b)
    if a.act == 1: ref = torch.nn.functional.gelu(ref)
    print("max abs err", (out.float() - ref).abs().max().item())
