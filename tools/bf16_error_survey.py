"""Measured bf16-path errors against the CPU oracle (fp32) on the parity tests' shapes: max |logit - ref| / max |ref|
and the largest softmax-probability difference over the OA-DM draw range, per configuration.  Used to set the stated
bf16 tolerances of tests/test_gpu_parity.py from measurements instead of a guess."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import bytenet_ref as R  # noqa: E402
from evodiff_amd.model import ByteNetLMTime  # noqa: E402

dev = torch.device("cuda", 0)
CFGS = {"TINY": dict(d_embed=8, d_model=64, n_layers=8, kernel_size=5, r=128, slim=True, activation="gelu"),
        "MID": dict(d_embed=8, d_model=384, n_layers=10, kernel_size=5, r=128, slim=True, activation="gelu"),
        "BIG512": dict(d_embed=8, d_model=512, n_layers=9, kernel_size=5, r=128, slim=True, activation="gelu"),
        "BIG640": dict(d_embed=8, d_model=640, n_layers=8, kernel_size=5, r=128, slim=False, activation="gelu"),
        "38M": R.MODEL_CONFIGS["38M"], "640M": R.MODEL_CONFIGS["640M"]}
CASES = [("TINY", 2, 100), ("MID", 2, 300), ("BIG512", 9, 300), ("BIG640", 16, 130), ("BIG640", 4, 512), ("38M", 2, 100),
         ("38M", 4, 1024), ("640M", 2, 256), ("640M", 2, 512)]
for name, B, L in CASES:
    cfg = CFGS[name]
    for scale in (1.0, 8.0):
        oracle = R.build_model(cfg, None, 0)
        with torch.no_grad():
            oracle.decoder.conv.weight.mul_(scale)
        m = ByteNetLMTime(31, cfg["d_embed"], cfg["d_model"], cfg["n_layers"], 5, 128, padding_idx=R.MASK_ID, final_ln=True,
                          slim=cfg["slim"], activation="gelu").eval()
        m.load_state_dict(oracle.state_dict())
        rng = np.random.default_rng(B * 1000 + L)
        x = torch.from_numpy(rng.integers(0, 26, (B, L)))
        x[:, rng.permutation(L)[: L // 2]] = R.MASK_ID
        with torch.no_grad():
            ref = oracle(x, torch.zeros(B, dtype=torch.long))
        t = torch.zeros(B, dtype=torch.long, device=dev)
        out = {}
        for tiles in (130, 0):
            m.precision = "bf16"
            m.set_option("min_tuned_tiles", tiles)
            got = m(x.to(dev), t).cpu()
            rel = float((got - ref).abs().max() / ref.abs().max())
            dp = float((torch.softmax(got[..., :21], -1) - torch.softmax(ref[..., :21], -1)).abs().max())
            out[tiles] = (rel, dp)
        print(f"{name:7s} B={B:3d} L={L:5d} decoder x{scale:3.0f}: default kernels rel {out[130][0]:.2e} dP {out[130][1]:.2e} | "
              f"tuned everywhere rel {out[0][0]:.2e} dP {out[0][1]:.2e}", flush=True)
        m.release()
