#!/usr/bin/env python3
"""Micro-benchmark of one implicit-GEMM problem of the ByteNet block through the C ABI (ed_debug_gemm).

    python tools/gemm_bench.py [--B 64 --L 256 --N 1280 --K 1280 --taps 5 --dil 1 --kernel 0 --iters 50]

Random bf16 operands (uniform [-1, 1): zero-filled operands read high on MI355X), zero halo rows for the
convolution, HIP-event timing over `iters` back-to-back launches.  Prints TFLOP/s on in-range taps only.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evodiff_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    for name, default in (("B", 64), ("L", 256), ("N", 1280), ("K", 1280), ("taps", 5), ("dil", 1), ("kernel", 0),
                          ("iters", 50), ("flat", 0), ("resid", 0), ("out_bf16", 0), ("check", 0)):
        ap.add_argument("--" + name, type=int, default=default)
    a = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    B, L = (1, a.B * a.L) if a.flat else (a.B, a.L)
    halo = (a.taps // 2) * a.dil
    pitch = halo + ((L + 255) // 256) * 256 + halo
    A = torch.zeros((B, pitch, a.K), dtype=torch.bfloat16, device=dev)
    A[:, halo:halo + L] = (torch.rand((B, L, a.K), device=dev) * 2 - 1).to(torch.bfloat16)
    W = ((torch.rand((a.taps, a.N, a.K), device=dev) * 2 - 1) / (a.K ** 0.5)).to(torch.bfloat16)
    bias = torch.rand(a.N, device=dev)
    out = torch.empty((B * L, a.N), dtype=torch.bfloat16 if a.out_bf16 else torch.float32, device=dev)
    resid = torch.rand((B * L, a.N), device=dev) if a.resid else None
    stream = torch.cuda.current_stream().cuda_stream

    def launch():
        _lib.check(lib.ed_debug_gemm(A.data_ptr(), pitch, halo, W.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                     resid.data_ptr() if a.resid else None, B, L, a.N, a.K, a.taps, a.dil, 1,
                                     a.out_bf16, a.kernel, stream))

    for _ in range(5):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    rows = sum(max(0, L - abs(t - a.taps // 2) * a.dil) for t in range(a.taps))
    flops = 2.0 * B * rows * a.N * a.K
    print(f"B={B} L={L} N={a.N} K={a.K} taps={a.taps} dil={a.dil} kernel={a.kernel}: {ms * 1e3:.1f} us/launch, "
          f"{flops / ms / 1e9:.1f} TFLOP/s (in-range taps)")
    if a.check:
        ref = torch.zeros((B, L, a.N), dtype=torch.float32, device=dev)
        Af = A.float()
        for t in range(a.taps):
            off = (t - a.taps // 2) * a.dil
            ref += Af[:, halo + off:halo + off + L] @ W[t].float().t()
        ref = ref.reshape(B * L, a.N) + bias
        if a.resid:
            pass  # resid aliasing is exercised by the model tests
        err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
        print("max rel err vs torch fp32:", err)


if __name__ == "__main__":
    main()
