"""A few eager OA-DM steps of a 640M-width model with TRACE_LAYERS blocks (default 8: 184 MB of bf16 weights, which
stay resident in the 256 MB Infinity Cache between steps, against 1.29 GB for the 56 blocks of OA_DM_640M); meant to
run under rocprofv3 --kernel-trace to compare per-launch durations on warm and on cold weights."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from evodiff_amd.model import ByteNetLMTime
torch.manual_seed(0)
nl = int(os.environ.get("TRACE_LAYERS", "8"))
m = ByteNetLMTime(31, 8, 1280, nl, 5, 128, padding_idx=28, final_ln=True, slim=False, activation="gelu").eval()
m.precision = "bf16"
m.use_graph = False
m.set_option("prune", 0)
m.set_option("incremental", 0)
dev = torch.device("cuda", 0)
B, L, n = int(os.environ.get("TRACE_B", "1")), 256, 6
order = torch.from_numpy(np.random.default_rng(0).permutation(L)[:n].astype(np.int32))
uni = torch.rand((n, B), device=dev)
sample = torch.full((B, L), 28, dtype=torch.long, device=dev)
m.oadm_sample(sample, order, uni, 21)
torch.cuda.synchronize()
