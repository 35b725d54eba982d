"""Same-process A/B of an execution option: ms per diffusion step with the option at each of the given values,
alternating, per configuration (bf16, graph replay).
    python tools/opt_ab.py fuse_ln 0 1 [--pre other_option=value ...] [--cfg cfg2 cfg3 cfg5 ...]   (--cfg last)"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from evodiff_amd import pretrained  # noqa: E402

CASES = {"cfg1": ("38M", 1, 100, 100, "oadm"), "cfg2": ("640M", 64, 256, 32, "oadm"), "cfg3": ("640M", 64, 256, 12, "d3pm"),
         "cfg4": ("640M", 64, 512, 16, "oadm"), "cfg5": ("38M", 1024, 1024, 6, "oadm"), "b128": ("640M", 128, 256, 16, "oadm"),
         "b32": ("640M", 32, 256, 32, "oadm"), "b16": ("640M", 16, 256, 32, "oadm")}
args = sys.argv[1:]
cfgs = ["cfg2"]
if "--cfg" in args:
    k = args.index("--cfg")
    cfgs, args = args[k + 1:], args[:k]
pre = {}
if "--pre" in args:                       # other options held fixed: --pre name=value name=value
    k = args.index("--pre")
    pre = dict((a.split("=")[0], int(a.split("=")[1])) for a in args[k + 1:])
    args = args[:k]
option, values = args[0], [int(v) for v in args[1:]]
dev = torch.device("cuda", 0)
for name in cfgs:
    size, B, L, n, sampler = CASES[name]
    torch.manual_seed(0)
    if sampler == "d3pm":
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):
            m, _, tok, _, T, Qb, Q = pretrained.D3PM_BLOSUM_640M(return_all=True, random_init=True)
        Q, Qb = Q.to(dev), Qb.to(dev)
        uni = torch.rand((T, B, L), device=dev)
    else:
        m, _, tok, _ = (pretrained.OA_DM_640M if size == "640M" else pretrained.OA_DM_38M)(random_init=True)
        order = torch.from_numpy(np.random.default_rng(0).permutation(L)[:n].astype(np.int32))
        uni = torch.rand((n, B), device=dev)
    m.precision = "bf16"
    for name_, value_ in pre.items():
        m.set_option(name_, value_)
    best = {v: 1e9 for v in values}
    for rep in range(3):
        for v in values:
            m.set_option(option, v)
            for inner in range(2):
                if sampler == "d3pm":
                    sample = torch.randint(0, tok.K, (B, L), device=dev)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    m.d3pm_sample(sample, Q, Qb, uni, tok.K, t_start=T - 1, n_steps=n)
                else:
                    sample = torch.full((B, L), tok.mask_id, dtype=torch.long, device=dev)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    m.oadm_sample(sample, order, uni, 21)
                torch.cuda.synchronize()
                if inner:
                    best[v] = min(best[v], (time.perf_counter() - t0) / n)
    print(f"{name} ({size} B={B} L={L} {sampler}): " + ", ".join(f"{option}={v}: {best[v] * 1e3:.3f} ms/step" for v in values), flush=True)
    m.release()
    del m
    torch.cuda.empty_cache()
