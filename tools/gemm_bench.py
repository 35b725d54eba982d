#!/usr/bin/env python3
"""Micro-benchmark of one implicit-GEMM problem of the ByteNet block through the C ABI (ed_debug_gemm).

    python tools/gemm_bench.py [--B 64 --L 256 --N 1280 --K 1280 --taps 5 --dil 1 --kernel 0 --iters 50]

--kernel: 0 library's choice, 1 generic, 2 tuned, 3 tuned without MFMAs, 4 tuned without copies, 5 per-tap form of
the convolution, 6/7 other copy placements of the windowed convolution, 8 staggered copies, 9 in-kernel clock stamps.
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
    A = torch.zeros((B * pitch + 8, a.K), dtype=torch.bfloat16, device=dev)[:B * pitch].view(B, pitch, a.K)  # + over-read slack
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

    stamps = None
    if a.kernel >= 100:
        tile, ks = (64, a.kernel - 200) if a.kernel >= 200 else (128, a.kernel - 100)
        wgs = ((a.N + tile - 1) // tile) * B * ((L + tile - 1) // tile) * ks
        stamps = torch.zeros(4 * wgs, dtype=torch.int64, device=dev)
        _lib.check(lib.ed_debug_set_stamp_buffer(stamps.data_ptr(), stamps.numel()))
    elif a.kernel == 9:
        stamps = torch.zeros(8 * B * ((L + 255) // 256) * ((a.N + 255) // 256), dtype=torch.int64, device=dev)
        _lib.check(lib.ed_debug_set_stamp_buffer(stamps.data_ptr(), stamps.numel()))
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
    if stamps is not None and a.kernel >= 100:
        st = stamps.view(-1, 4).cpu().double() / 100.0     # us; the last launch's values
        t0 = st[:, 0].min()
        q = lambda v: f"{v.min().item():.2f}/{v.median().item():.2f}/{v.max().item():.2f}"
        print(f"stamps over {len(st)} workgroups (us after the first workgroup's entry, min/median/max): entry {q(st[:, 0] - t0)}, "
              f"first operands in LDS {q(st[:, 1] - t0)}, K loop done {q(st[:, 2] - t0)}, stores done {q(st[:, 3] - t0)}")
        _lib.check(lib.ed_debug_set_stamp_buffer(None, 0))
    elif stamps is not None:
        st = stamps.view(-1, 8).cpu().double()[:, :4]
        st = st[st[:, 3] > 0]
        clk = (st[:, :3].sum(1) / st[:, 3]).median().item() * 100.0
        steps = a.taps * (a.K // 64)
        med = st.median(0).values
        print(f"stamps (median over {len(st)} tiles): prologue {med[0]:.0f}, K loop {med[1]:.0f} ({med[1] / steps:.0f} per "
              f"step), epilogue {med[2]:.0f} shader cycles; in-kernel clock {clk:.0f} MHz; stamped span "
              f"{med[3] / 100.0:.1f} us (max {st[:, 3].max().item() / 100.0:.1f})")
        _lib.check(lib.ed_debug_set_stamp_buffer(None, 0))
    if a.check:
        ref = torch.zeros((B, L, a.N), dtype=torch.float32, device=dev)
        Af = A.float()
        for t in range(a.taps):
            off = (t - a.taps // 2) * a.dil
            ref += Af[:, halo + off:halo + off + L] @ W[t].float().t()
        ref = ref.reshape(B * L, a.N) + bias
        if a.resid:
            ref += resid
        err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
        print("max rel err vs torch fp32:", err)


if __name__ == "__main__":
    main()
