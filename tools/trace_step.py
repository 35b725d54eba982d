"""Runs a few eager OA-DM steps (640M, B=64, L=256, bf16); meant to run under rocprofv3 --kernel-trace."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from evodiff_amd.pretrained import OA_DM_640M
torch.manual_seed(0)
m, _, tok, _ = OA_DM_640M(random_init=True)
m.precision = "bf16"
m.use_graph = False
dev = torch.device("cuda", 0)
import os
B, L, n = int(os.environ.get("TRACE_B", "64")), 256, 4
order = torch.from_numpy(np.random.default_rng(0).permutation(L)[:n].astype(np.int32))
uni = torch.rand((n, B), device=dev)
sample = torch.full((B, L), tok.mask_id, dtype=torch.long, device=dev)
m.oadm_sample(sample, order, uni, 21)
torch.cuda.synchronize()
