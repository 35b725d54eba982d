"""Multi-GPU sampling: independent sequences are sharded across ranks (one process per GPU).

The reference samples on a single device (generate.py:78-80).  Sequences are independent given the weights,
so rank r of W samples the contiguous shard ``shard_range(N, r, W)`` with its own full weight replica, its own
position order / uniform streams (seeded by GLOBAL sequence index, so results do not depend on W) and its own
captured graph.  There is no collective on the data path; the one exchange is the all-gather of the finished
token tensors (RCCL over xGMI when the backend is "nccl"; gloo on CPU in the tests).
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_sequences, rank, world_size):
    """Contiguous [lo, hi) of sequence indices for ``rank``; sizes differ by at most one."""
    base, extra = divmod(n_sequences, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_seeds(seed, lo, hi):
    """One RNG seed per global sequence index."""
    return [seed * 1000003 + i for i in range(lo, hi)]


def global_sub_batches(lo, hi, batch):
    """Cut the shard [lo, hi) along the GLOBAL grid of sub-batches: sub-batch j holds the sequences
    [j * batch, (j + 1) * batch) whatever the number of ranks, and a rank samples the part of each that falls into its
    shard, ``(a, b, j)`` with ``[a, b)`` inside sub-batch j.  One sampler call per piece: the position order is one
    per call in the reference (generate.py:191-192), so it is seeded by j and every sequence sees the same order at any
    world size, also when the shard size is not a multiple of ``batch``."""
    pieces = []
    a = lo
    while a < hi:
        j = a // batch
        b = min(hi, (j + 1) * batch)
        pieces.append((a, b, j))
        a = b
    return pieces


def batch_order(seed, j, length):
    """The position order of global sub-batch j (np.random.shuffle of arange(L) in the reference, generate.py:191-192)."""
    return np.random.default_rng([int(seed), int(j)]).permutation(length).astype(np.int32)


def sequence_uniforms(seed, lo, hi, shape):
    """Pre-drawn uniforms of the sequences [lo, hi): one stream per GLOBAL sequence index, so a sequence draws the same
    numbers on any rank of any world size.  Returns fp32 ``shape + (hi - lo,)`` (sequence index last: OA-DM wants
    (n_steps, B)); D3PM callers move the axis."""
    shape = tuple(shape)
    out = np.empty(shape + (hi - lo,), dtype=np.float32)
    for g in range(lo, hi):
        out[..., g - lo] = np.random.default_rng([int(seed), int(g)]).random(shape, dtype=np.float32)
    return out


def gather_sequences(local_tokens, n_sequences):
    """All-gather the (n_local, L) token tensors of every rank into (n_sequences, L), in global order."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_tokens
    world = dist.get_world_size()
    n_max = -(-n_sequences // world)
    L = local_tokens.shape[1]
    padded = torch.zeros((n_max, L), dtype=local_tokens.dtype, device=local_tokens.device)
    padded[: local_tokens.shape[0]] = local_tokens
    out = torch.empty((world * n_max, L), dtype=local_tokens.dtype, device=local_tokens.device)
    dist.all_gather_into_tensor(out, padded)
    pieces = []
    for r in range(world):
        lo, hi = shard_range(n_sequences, r, world)
        pieces.append(out[r * n_max: r * n_max + (hi - lo)])
    return torch.cat(pieces, dim=0)
