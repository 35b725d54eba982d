#!/usr/bin/env python3
"""Where the time of the three fused block kernels goes, from in-kernel 100-MHz clock stamps (ed_debug_gemm 500).

    python tools/kernel_timeline.py [--B 64 --L 256 --C 1280 --iters 20 --unfused 0]

Launches PFF1 -> convolution -> PFF2 (640M geometry, LayerNorm + GELU in each epilogue) back to back, as a block of
the denoiser does, and stamps the last round plus the PFF1 behind it: per kernel the spread of workgroup entries, the
prologue, K loop and epilogue of a workgroup (median / max), the spread of exits, and the gap between the last exit of
one launch and the first entry of the next.  All times in microseconds.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evodiff_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    for name, default in (("B", 64), ("L", 256), ("C", 1280), ("iters", 20), ("unfused", 0), ("dil", 1), ("dbg", 0), ("xcd", 0), ("streams", 1)):
        ap.add_argument("--" + name, type=int, default=default)
    a = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    B, L, C = a.B, a.L, a.C
    halo = 2 * a.dil
    pitch = halo + ((L + 255) // 256) * 256 + halo
    rnd = lambda *shape: (torch.rand(shape, device=dev) * 2 - 1)
    # activations as the block sees them: GELU of a normalised row (the chip's clock depends on the operands' bits)
    act = lambda *shape: torch.nn.functional.gelu(torch.randn(shape, device=dev)).to(torch.bfloat16)
    A1 = act(B * L + 8, C)
    Ac = torch.zeros((B * pitch + 8, C), dtype=torch.bfloat16, device=dev)
    Ac[:B * pitch].view(B, pitch, C)[:, halo:halo + L] = act(B, L, C)
    W1 = (rnd(1, C, C) / C ** 0.5).to(torch.bfloat16)
    Wc = (rnd(5, C, C) / (5 * C) ** 0.5).to(torch.bfloat16)
    W2 = (rnd(1, C, C) / C ** 0.5).to(torch.bfloat16)
    bias = torch.rand(C, device=dev)
    out16 = torch.empty((B * L, C), dtype=torch.bfloat16, device=dev)
    out32 = torch.empty((B * L, C), dtype=torch.float32, device=dev)
    resid = torch.rand((B * L, C), device=dev)
    # --streams 2: consecutive launches alternate between two streams with nothing ordering them (timing experiment:
    # how much of a launch's tail, the gap behind it and the next prologue would overlap; the data are independent here)
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(a.streams - 1)]
    turn = [0]

    def next_stream():
        turn[0] += 1
        return streams[turn[0] % len(streams)].cuda_stream
    tiles = B * ((L + 255) // 256) * ((C + 255) // 256)
    kern = 9 if a.unfused else 500 + a.dbg  # dbg: diagnostic bits of the fused epilogue (ln_epilogue.hpp)

    def pff1():
        _lib.check(lib.ed_debug_gemm(A1.data_ptr(), L, 0, W1.data_ptr(), bias.data_ptr(), out16.data_ptr(), None,
                                     B, L, C, C, 1, 1, 1, 1, kern, next_stream()))

    def conv():
        _lib.check(lib.ed_debug_gemm(Ac.data_ptr(), pitch, halo, Wc.data_ptr(), bias.data_ptr(), out16.data_ptr(), None,
                                     B, L, C, C, 5, a.dil, 1, 1, kern, next_stream()))

    def pff2():
        _lib.check(lib.ed_debug_gemm(A1.data_ptr(), L, 0, W2.data_ptr(), bias.data_ptr(), out32.data_ptr(),
                                     resid.data_ptr(), B, L, C, C, 1, 1, 1, 0, kern, next_stream()))

    names = ["PFF1", "conv", "PFF2", "PFF1'"]
    fns = [pff1, conv, pff2, pff1]
    bufs = [torch.zeros(8 * tiles, dtype=torch.int64, device=dev) for _ in range(5)]
    _lib.check(lib.ed_debug_set_stamp_buffer(bufs[4].data_ptr(), bufs[4].numel()))
    pff1(); conv(); pff2()      # (first calls allocate the diagnostic buffers)
    torch.cuda.synchronize()
    for _ in range(a.iters):
        pff1(); conv(); pff2()
    for i, fn in enumerate(fns):
        _lib.check(lib.ed_debug_set_stamp_buffer(bufs[i].data_ptr(), bufs[i].numel()))
        fn()
    torch.cuda.synchronize()
    _lib.check(lib.ed_debug_set_stamp_buffer(None, 0))
    st = [b.view(-1, 8).cpu().double() for b in bufs[:4]]
    st = [s[s[:, 4] > 0] for s in st]  # (the buffers are sized for 256-column tiles; 320-column tiles use fewer slots)
    t0 = st[0][:, 4].min()
    us = lambda v: (v - t0) / 100.0
    print(f"B={B} L={L} C={C} dilation={a.dil} {'unfused' if a.unfused else 'LayerNorm fused'}; {len(st[0])} workgroups per launch; "
          "us after the first entry of PFF1")
    prev_exit = None
    for n, s in zip(names, st):
        entry, l0, l1, ex = us(s[:, 4]), us(s[:, 5]), us(s[:, 6]), us(s[:, 7])
        if n == "conv":  # the windowed convolution stamps its loop's duration; slot 6 is 'statistics published' there
            l1 = l0 + s[:, 3] / 100.0
        q = lambda v: f"{v.median().item():6.1f} (max {v.max().item():6.1f})"
        gap = "" if prev_exit is None else f"  gap after previous launch {entry.min().item() - prev_exit:5.1f}"
        print(f"{n:6s} entry {entry.min().item():7.1f} .. {entry.max().item():7.1f} | prologue {q(l0 - entry)} | K loop {q(l1 - l0)} "
              f"| epilogue {q(ex - l1)} | exit {ex.min().item():7.1f} .. {ex.max().item():7.1f} | first entry -> last exit "
              f"{ex.max().item() - entry.min().item():6.1f}{gap}")
        prev_exit = ex.max().item()
        if n == "conv" and not a.unfused:  # finer stamps of the fused epilogue
            pub, arr, stat = us(s[:, 6]), us(s[:, 0]), us(s[:, 2])
            print(f"       epilogue of the convolution: statistics published {q(pub - l1)} | partners arrived {q(arr - pub)} | "
                  f"row statistics ready {q(stat - arr)} | normalise + GELU + stores drained {q(ex - stat)}")
        if a.xcd and len(s) % 8 == 0:  # tiles [x*q, (x+1)*q) run on XCD x (the kernels' tile order)
            qn = len(s) // 8
            loop, rel = (l1 - l0).view(8, qn), (ex - entry.min()).view(8, qn)
            print("       per XCD: K loop median " + " ".join(f"{v:6.1f}" for v in loop.median(1).values.tolist()) +
                  " | last exit " + " ".join(f"{v:6.1f}" for v in rel.max(1).values.tolist()))
    period = us(st[3][:, 4]).min().item()
    print(f"block period (first entry of PFF1 -> first entry of the next PFF1): {period:.1f} us")


if __name__ == "__main__":
    main()
