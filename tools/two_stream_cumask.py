#!/usr/bin/env python3
"""Experiment: two half-batches of OA-DM sampling on two HIP streams, one host thread per stream (a single thread blocks
on the first stream's full queue), optionally with each stream restricted to half of the CUs of every XCD.

usage: two_stream_cumask.py MODE
  MODE: one (B = 64, one stream) | threads (2 x 32, plain streams) | mask (2 x 32, disjoint CU halves per XCD)
"""
import ctypes
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evodiff_amd import pretrained  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "mask"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
B, L = 64, 256
nsplit = 1 if mode == "one" else 2
Bs = B // nsplit

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(words):
    s = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(len(words)), arr)
    assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
    return torch.cuda.ExternalStream(s.value, device=dev)


if mode == "mask":
    # bit i of the mask = XCD (i % 8), CU slot (i / 8): the first 128 bits are CU slots 0..15 of every XCD
    streams = [masked_stream([0xFFFFFFFF] * 4 + [0] * 4), masked_stream([0] * 4 + [0xFFFFFFFF] * 4)]
elif mode == "threads":
    streams = [torch.cuda.Stream() for _ in range(2)]
else:
    streams = [torch.cuda.Stream()]

models = []
for i in range(nsplit):
    torch.manual_seed(0)
    m, _, tok, _ = pretrained.OA_DM_640M(random_init=True)
    m.precision = "bf16"
    m.hip_handle(dev)
    models.append(m)

rng = np.random.default_rng(0)
order = torch.from_numpy(rng.permutation(L).astype(np.int32)).to(dev)
uni = torch.rand((L, Bs), device=dev)


def worker(m, s, x):
    torch.cuda.set_device(0)
    with torch.cuda.stream(s):
        m.oadm_sample(x, order, uni, 21)
    s.synchronize()


def run():
    samples = [torch.full((Bs, L), tok.mask_id, dtype=torch.long, device=dev) for _ in range(nsplit)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=a) for a in zip(models, streams, samples)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert all(int((x == tok.mask_id).sum()) == 0 for x in samples)
    return dt, samples


for a in zip(models, streams, [torch.full((Bs, L), tok.mask_id, dtype=torch.long, device=dev) for _ in range(nsplit)]):
    worker(*a)          # graph capture, one handle after the other
run()
for _ in range(2):
    dt, samples = run()
    print(f"mode={mode}: {dt:.3f} s per {B} sequences -> {B / dt:.2f} seq/s", flush=True)
if nsplit == 2:
    print("halves equal:", bool((samples[0] == samples[1]).all()), flush=True)
