#!/usr/bin/env python3
"""Block chain against single launches on the 640M block geometry: where do the logits differ?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import bytenet_ref as R  # noqa: E402
from evodiff_amd.model import ByteNetLMTime  # noqa: E402

n_layers = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = dict(d_embed=8, d_model=1280, n_layers=n_layers, kernel_size=5, r=128, slim=False, activation="gelu")
oracle = R.build_model(cfg, None, 77)
dev = torch.device("cuda", 0)
m = ByteNetLMTime(len(R.MSA_ALPHABET), cfg["d_embed"], cfg["d_model"], cfg["n_layers"], cfg["kernel_size"], cfg["r"],
                  causal=False, padding_idx=R.MASK_ID, rank=None, dropout=0.0, tie_weights=False, final_ln=True,
                  slim=cfg["slim"], activation=cfg["activation"], timesteps=None)
m.load_state_dict(oracle.state_dict())
m.precision = "bf16"
m = m.eval()
B, L = 64, 256
x = torch.from_numpy(np.random.default_rng(5).integers(0, 31, (B, L))).to(dev)
t = torch.zeros(B, dtype=torch.long, device=dev)
m.set_option("block_chain", 0)
single = m(x, t).float().cpu().numpy()
for blocks in (1, 1, 2):
    m.set_option("block_chain", blocks)
    out = m(x, t).float().cpu().numpy()
    bad = ~np.isclose(out, single, rtol=0, atol=0, equal_nan=False)
    rows = bad.any(-1)
    print(f"chain {blocks}: {rows.sum()} of {rows.size} rows differ; nan rows {np.isnan(out).any(-1).sum()}; "
          f"sequences with a difference: {np.flatnonzero(rows.any(1))[:20]} ...; max abs diff {np.nanmax(np.abs(out - single)):.3g}")
    if rows.any():
        b = np.flatnonzero(rows.any(1))[0]
        print("   first bad sequence", b, "bad positions", np.flatnonzero(rows[b])[:32])
try:
    m.check_errors()
except Exception as e:
    print("async error:", e)
