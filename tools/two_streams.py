"""A/B: one OA-DM batch of 64 on one stream against two half batches of 32 on two streams (two host threads, two models
with the same weights), and against two batches of 64 on two streams.  Question: does letting the workgroups of
different launches fall out of step (K loops of one launch next to the HBM-bound epilogues of another) raise the
throughput of a power-limited chip?  Prints seconds per 64 sequences for each form."""
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from evodiff_amd import pretrained  # noqa: E402

dev = torch.device("cuda", 0)
L = 256


def make():
    torch.manual_seed(0)
    m, _, tok, _ = pretrained.OA_DM_640M(random_init=True)
    m.precision = "bf16"
    m.use_graph = True
    m.set_option("min_tuned_tiles", 100)
    m.hip_handle(dev)
    return m, tok


def job(m, tok, B, seed, stream, out):
    rng = np.random.default_rng(seed)
    order = torch.from_numpy(rng.permutation(L).astype(np.int32)).to(dev)
    uni = torch.from_numpy(rng.random((L, B), dtype=np.float32)).to(dev)
    sample = torch.full((B, L), tok.mask_id, dtype=torch.long, device=dev)
    torch.cuda.synchronize()

    def run():
        with torch.cuda.stream(stream):
            m.oadm_sample(sample, order, uni, len(tok.all_aas) - 6)
        out.append(sample)
    return run


def timed(runs):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=r) for r in runs]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


ma, tok = make()
mb, _ = make()
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
out = []
for name, runs, seqs in (
        ("1 x 64 on one stream", lambda: [job(ma, tok, 64, 1, sa, out)], 64),
        ("2 x 32 on two streams", lambda: [job(ma, tok, 32, 1, sa, out), job(mb, tok, 32, 2, sb, out)], 64),
        ("2 x 64 on two streams", lambda: [job(ma, tok, 64, 1, sa, out), job(mb, tok, 64, 2, sb, out)], 128),
        ("1 x 32 on one stream", lambda: [job(ma, tok, 32, 1, sa, out)], 32)):
    timed(runs())                                     # capture + warm-up
    ts = [timed(runs()) for _ in range(2)]
    print("%-24s %.3f s per 64 sequences (runs: %s)" % (name, min(ts) * 64 / seqs, ", ".join("%.3f" % t for t in ts)),
          flush=True)
ma.check_errors()
mb.check_errors()
