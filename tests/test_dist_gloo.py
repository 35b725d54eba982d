"""Multi-process path on CPU (gloo, world_size 2): sequences are sharded across ranks with no data-path
collective; the only exchange is the all-gather of finished token tensors.  The sharding and
gather logic of bench.py is exercised here with a stand-in sampler, since the HIP sampler needs a GPU."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from evodiff_amd.parallel import gather_sequences, shard_range, shard_seeds
    N, L = 10, 16                                   # uneven: 10 sequences over 2 ranks (5 + 5), then 3 ranks' math below
    lo, hi = shard_range(N, rank, world)
    seeds = shard_seeds(1234, lo, hi)
    # stand-in "sampler": tokens depend only on the global sequence index => independent of the world size
    local = torch.stack([torch.from_numpy(np.random.default_rng(s).integers(0, 21, L)) for s in seeds])
    full = gather_sequences(local, N)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sequence_sharding_and_gather(tmp_path):
    world, port = 2, 29500 + (os.getpid() % 1000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from evodiff_amd.parallel import shard_range, shard_seeds
    want = np.stack([np.random.default_rng(s).integers(0, 21, 16) for s in shard_seeds(1234, 0, 10)])
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npy")
        assert got.shape == (10, 16) and np.array_equal(got, want)      # identical on every rank, and == 1-rank result
    # ragged partition: every sequence exactly once, contiguous, sizes differ by at most one
    for n, w in ((10, 3), (4096, 8), (1, 4), (7, 8)):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1
