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


def _bench_line(*flags):
    """One `python bench.py --dry-run ...` run as the driver would start it (no torchrun around it): the stand-in
    sampler on the CPU, gloo between the ranks bench.py starts itself.  Returns the parsed JSON line."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", *flags], env=env,
                         capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, run.stdout                   # rank 0 prints ONE line, whatever the number of ranks
    return json.loads(lines[0])


def test_bench_self_launch_and_world_size_independent_tokens():
    """bench.py's own code path: `python bench.py --gpus 2` starts two ranks itself instead of
    exiting, and orders / uniforms are seeded by global batch / sequence index, so the gathered tokens are identical at
    world 1 and 2 - in the strong form also when the shard size (11 -> 6 + 5) is not a multiple of --batch (4), and in
    the weak form for the same global batches."""
    base = ["--batch", "4", "--seq-len", "16", "--warmup", "0"]
    one = _bench_line("--gpus", "1", "--steps", "1", "--total-sequences", "11", *base)
    two = _bench_line("--gpus", "2", "--steps", "1", "--total-sequences", "11", *base)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and one["scaling"] == two["scaling"] == "strong"
    assert one["tokens_sha256"] == two["tokens_sha256"]
    other = _bench_line("--gpus", "2", "--steps", "1", "--total-sequences", "12", *base)
    assert other["tokens_sha256"] != one["tokens_sha256"]             # (the digest does see the tokens)
    weak1 = _bench_line("--gpus", "1", "--steps", "2", *base)        # global batches 0 and 1, one after the other
    weak2 = _bench_line("--gpus", "2", "--steps", "1", *base)        # global batches 0 and 1, side by side
    assert weak2["n_gpus"] == 2 and weak2["config"]["global_batch"] == 8 and weak2["scaling"] == "weak"
    assert weak1["tokens_sha256"] == weak2["tokens_sha256"]


def test_config4_strong_scaling_form_at_world_two():
    """BASELINE config 4's own flags (`--total-sequences ... --seq-len 512`, sub-batches of 64) through bench.py's
    launcher at world 1 and 2, with a sequence count that leaves the two shards uneven and not a multiple of the
    sub-batch (4095 -> 2048 + 2047): one line, n_gpus / rccl_ranks from the process group, equal tokens."""
    flags = ["--steps", "1", "--warmup", "0", "--total-sequences", "4095", "--seq-len", "512", "--batch", "64"]
    one = _bench_line("--gpus", "1", *flags)
    two = _bench_line("--gpus", "2", *flags)
    assert (one["n_gpus"], one["rccl_ranks"]) == (1, 1) and (two["n_gpus"], two["rccl_ranks"]) == (2, 2)
    assert one["config"]["global_batch"] == two["config"]["global_batch"] == 4095 and two["scaling"] == "strong"
    assert one["tokens_sha256"] == two["tokens_sha256"]


def test_bench_refuses_a_rank_without_a_gpu():
    """`--gpus N` on a node with fewer GPUs must fail loudly on the rank that has none, before any collective: here (no
    GPU at all) already rank 0 of a one-rank job says so."""
    import subprocess
    if torch.cuda.device_count() > 0:
        return
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert run.returncode != 0 and "sees 0 GPU(s)" in run.stderr


def test_global_sub_batch_grid_and_streams():
    sys.path.insert(0, ROOT)
    from evodiff_amd.parallel import batch_order, global_sub_batches, sequence_uniforms, shard_range
    # a shard's pieces follow the global grid: never across a multiple of the batch, every sequence once
    for n, w, bs in ((11, 2, 4), (4096, 8, 64), (10, 3, 4), (5, 4, 8)):
        seen = []
        for r in range(w):
            lo, hi = shard_range(n, r, w)
            for a, b, j in global_sub_batches(lo, hi, bs):
                assert lo <= a < b <= hi and a // bs == j == (b - 1) // bs
                seen.extend(range(a, b))
        assert seen == list(range(n))
    u = sequence_uniforms(7, 3, 9, (5,))
    assert u.shape == (5, 6) and u.dtype == np.float32 and 0 <= u.min() and u.max() < 1
    assert np.array_equal(u[:, 2:4], sequence_uniforms(7, 5, 7, (5,)))      # a sequence's stream ignores its neighbours
    assert sorted(batch_order(1, 3, 16).tolist()) == list(range(16))
    assert not np.array_equal(batch_order(1, 3, 16), batch_order(1, 4, 16))
