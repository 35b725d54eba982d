#!/usr/bin/env python3
"""Throughput of the hot path on MI355X: OA_DM_640M unconditional sampling, length 256 (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

With ``--gpus N`` (N > 1) and no ``WORLD_SIZE`` in the environment the script starts its N ranks itself (a
``torch.distributed.run`` child process, before anything here touches the GPU), as the reference's own multi-process
entry does (train.py:75,84-88 ``mp.spawn``); under torchrun it is one rank of the job.

One "step" = one full pass of the hot path over one batch: ``generate_oaardm``'s loop for B=64 sequences of
L=256 (256 denoiser forwards + draws) on every rank, followed by the all-gather of the finished sequences when
N > 1.  Weights are synthetic (reference-constructor default init under seed 0), inputs are resident in HBM
before the timed region.  Rank 0 prints ONE JSON line.

Seeding.  Sequence g of the job (rank r samples g in [r*B, (r+1)*B) per step) draws its uniforms from a stream seeded
by g, and global batch j = g // B its position order from a stream seeded by j (evodiff_amd.parallel): the sampled
tokens do not depend on the number of ranks (``tokens_sha256`` in the line).

roofline: the dominant kernel is the windowed dilated convolution (conv_window_kernel: the convolution of every
block at L <= 256).  Its launches are timed live with HIP-event pairs on the launch stream on every 64th diffusion
step inside the timed region; achieved = algorithmic FLOPs of those launches (in-range taps only) /
their summed duration, against the 2.5 PFLOP/s dense bf16 MFMA peak.  traffic = memory-side bytes of one such launch
from the committed PMC passes (profiles/, tools/pmc_traffic.sh), for the default config only.  cpu_baseline: the CPU
oracle (a port of the reference's PyTorch path) timed on this host for a bounded sample, plus BASELINE config 1
(OA_DM_38M, B=1, L=100, all 100 steps: the reference notebook's case) at all threads and at 8.

``--dry-run``: no GPU and no model - a stand-in sampler on the CPU (gloo when N > 1) that exercises exactly this
file's launching, sharding, seeding and gathering (tests/test_dist_gloo.py).
"""
import argparse
import contextlib
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA
PEAK_FP32_TFLOPS = 157.3
# what a bare v_mfma_f32_16x16x32_bf16 loop on RANDOM operands sustains on this chip (the clock is held down under load;
# tools/micro/mfma_peak.hip, profiles/r02_mfma_sustained_peak_microbench.txt); reported beside the nominal peak, never as it
SUSTAINED_BF16_TFLOPS = 1953.0
PEAK_HBM_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E
SEED_ORDER, SEED_UNIFORM, SEED_START = 1000, 2000, 3000
# The synthetic weights are the reference constructor's default init under seed 0 with the decoder's weight scaled by
# this factor: default-init logits are nearly flat, so every draw would sit within bf16 noise of a category boundary and
# a check of the sampled tokens against the oracle would be vacuous.  The decoder runs on one row per sequence and step
# (fp32, outside every GEMM): the factor changes which tokens are drawn, not the work.
DECODER_SCALE = 8.0


def sharpen_decoder(model, scale):
    with torch.no_grad():
        model.decoder.conv.weight.mul_(float(scale))


def oracle_spot_check(model, cfg_name, final, order, uniforms, precision, timeout_rows=3, n_steps_checked=12):
    """After the timed region: the sequences the LAST timed step sampled, checked against the CPU oracle draw by draw
    (oracle/spot_check.py: three rows x a dozen steps spread over the loop, the oracle teacher-forced along the GPU's own
    trajectory).  The oracle is the checker here, never the thing measured."""
    from oracle import bytenet_ref as R
    from oracle.spot_check import oadm_spot_check, spread_steps
    t0 = time.perf_counter()
    oracle = R.build_model(R.MODEL_CONFIGS[cfg_name], None, None)
    oracle.load_state_dict(model.state_dict())
    B, L = final.shape
    rows = sorted({0, B // 2, B - 1})[:timeout_rows]
    tol = 5e-2 if precision == "bf16" else 1e-4          # tests/test_gpu_parity.py: DRAW_MARGIN_BF16 / _FP32
    steps = spread_steps(len(order), n_steps_checked)
    res = oadm_spot_check(oracle, final, order.cpu().numpy(), uniforms, rows, steps, tol)
    res.update({"rows": rows, "steps": steps, "margin_tol": tol, "seconds": round(time.perf_counter() - t0, 1),
                "what": "oracle teacher-forced along the sampled trajectory of the last timed batch; a draw is compared "
                        "when its uniform is further than margin_tol (probability mass) from a category boundary"})
    return res


def _time_oadm_oracle(R, model, L, seconds_budget, min_steps, max_steps):
    rng = np.random.default_rng(0)
    order = rng.permutation(L)
    uni = torch.from_numpy(rng.random((L, 1), dtype=np.float32))
    sample = torch.full((1, L), R.MASK_ID, dtype=torch.long)
    sample, *_ = R.oadm_step(model, sample, int(order[0]), uni[0])          # warm-up
    t0 = time.perf_counter()
    n = 0
    while n < min_steps or (time.perf_counter() - t0 < seconds_budget and n < max_steps):
        sample, *_ = R.oadm_step(model, sample, int(order[(n + 1) % L]), uni[(n + 1) % L])
        n += 1
    return n, (time.perf_counter() - t0) / n


def cpu_baseline(L, seconds_budget=16.0):
    """The CPU oracle: (a) the benchmarked model and length at B=1 for a bounded number of diffusion steps (the value
    of the object), (b) BASELINE config 1 exactly - OA_DM_38M, B=1, L=100, all 100 steps (examples/evodiff.ipynb:160
    shows 9.80 it/s on a laptop CPU) - at every thread this process may use and at 8 threads."""
    from oracle import bytenet_ref as R
    threads = torch.get_num_threads()
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    model = R.build_model(R.MODEL_CONFIGS["640M"], None, 0)
    n, per_step = _time_oadm_oracle(R, model, L, seconds_budget, 8, L - 1)
    del model
    out = {"value": 1.0 / (per_step * L), "unit": "sequences/sec", "cores": threads, "kind": "port",
           "affinity_cores": affinity,
           "sample": f"oracle (fp32 PyTorch CPU) OA_DM_640M B=1 L={L}: {n} diffusion steps timed on {threads} threads "
                     f"({affinity} cores in this process's affinity mask), {per_step:.3f} s/step, extrapolated to {L} "
                     f"steps per sequence"}
    small = R.build_model(R.MODEL_CONFIGS["38M"], None, 0)
    legs = []
    for nt in sorted({threads, min(8, threads)}, reverse=True):
        torch.set_num_threads(nt)
        n1, ps = _time_oadm_oracle(R, small, 100, 1e9, 99, 99)             # every step of one sequence
        legs.append({"threads": nt, "it_per_s": 1.0 / ps, "sequences_per_s": 1.0 / (ps * 100), "steps_timed": n1 + 1})
    torch.set_num_threads(threads)
    out["config1"] = {"workload": "OA_DM_38M generate_oaardm B=1 L=100, all 100 steps (BASELINE config 1; reference "
                                  "notebook: 9.80 it/s on an unnamed laptop CPU)", "kind": "port", "legs": legs}
    return out


def f_eff_per_sequence(hp, L):
    """Algorithmic FLOPs of ONE forward over one sequence of length L (in-range taps only)."""
    D, H, ks = hp["d_model"], hp["d_hidden"], hp["kernel_size"]
    period = int(np.log2(hp["r"])) + 1
    total = 0.0
    for n in range(hp["n_layers"]):
        d = 2 ** (n % period)
        taps = sum(max(0, L - abs(o) * d) for o in range(-(ks // 2), ks // 2 + 1))
        total += L * 2.0 * (2 * D * H) + 2.0 * H * H * taps
    return total + L * 2.0 * (hp["d_embed"] * D + hp["n_tokens"] * D)


def measured_traffic(args):
    """Memory-side bytes per launch of the dominant kernel, from the committed PMC summary (None if the run's
    configuration is not the one the counters were collected on)."""
    if not (args.model == "640M" and args.batch == 64 and args.seq_len == 256 and args.precision == "bf16"):
        return None
    for name in ("r03_conv_window_traffic.json", "r02_conv_window_traffic.json", "r01_conv_window_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return json.load(f)["traffic_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            continue
    return None


def tokens_digest(tokens):
    return hashlib.sha256(np.ascontiguousarray(tokens.detach().cpu().numpy().astype(np.int64)).tobytes()).hexdigest()


def standin_oadm_sample(sample, order, uniforms, n_draw):
    """--dry-run: what the sampler's contract fixes, without a model - token (b, order[s]) is a function of
    uniforms[s, b] and of the step it was drawn at, so order, uniforms and their seeding all show in the result."""
    for s, pos in enumerate(order.tolist()):
        sample[:, pos] = ((uniforms[s] * n_draw).long() + s) % n_draw


class Job:
    """Process group, device and fences of one rank."""

    def __init__(self, args):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dry = args.dry_run
        if self.world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}")
        if not self.dry:
            # fail loudly and early, on every rank, before any collective can hang on a missing peer
            have = torch.cuda.device_count()
            if have < self.local_rank + 1:
                raise SystemExit(f"rank {self.rank}: LOCAL_RANK={self.local_rank} but this process sees {have} GPU(s); "
                                 f"a {self.world}-rank job needs one GPU per rank on this node")
        # the collectives run when there is more than one rank - or, to rehearse the RCCL calls on a one-GPU box, when a
        # launcher set up a one-rank group and EVODIFF_BENCH_FORCE_COLLECTIVE=1 asks for them
        self.collective = self.world > 1 or ("WORLD_SIZE" in os.environ and
                                             os.environ.get("EVODIFF_BENCH_FORCE_COLLECTIVE") == "1")
        if self.dry:
            self.dev = torch.device("cpu")
        else:
            torch.cuda.set_device(self.local_rank)
            self.dev = torch.device("cuda", self.local_rank)
        if self.collective:
            import torch.distributed as dist
            if self.dry:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=self.dev)
            if dist.get_world_size() != self.world or dist.get_rank() != self.rank:
                raise SystemExit(f"process group has {dist.get_world_size()} ranks (this is rank {dist.get_rank()}) but the "
                                 f"environment says WORLD_SIZE={self.world} RANK={self.rank}")

    def group_size(self):
        """Ranks that really take part: the process group's size when there is one."""
        if self.collective:
            import torch.distributed as dist
            return dist.get_world_size()
        return 1

    def fence(self):
        if not self.dry:
            torch.cuda.synchronize(self.dev)
        if self.collective:
            import torch.distributed as dist
            dist.barrier()
            if not self.dry:
                torch.cuda.synchronize(self.dev)

    def max_over_ranks(self, elapsed):
        if not self.collective:
            return elapsed
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=self.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.collective:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()


def strong_scaling(args, job, model, tokenizer_mask, n_draw):
    """BASELINE config 4's form: a fixed number of sequences, sharded over the ranks (contiguous shards, no collective
    on the data path), each rank sampling its shard along the GLOBAL grid of sub-batches; the finished sequences are
    all-gathered once per step.  One "step" = the whole job.  Orders are seeded by the global sub-batch index and
    uniforms by the global sequence index, so the sampled sequences do not depend on the number of ranks, whatever the
    shard size."""
    from evodiff_amd.parallel import (batch_order, gather_sequences, global_sub_batches, sequence_uniforms,
                                      shard_range)
    N, L, Bsub = args.total_sequences, args.seq_len, args.batch
    dev, world, rank = job.dev, job.world, job.rank
    lo, hi = shard_range(N, rank, world)
    subs = global_sub_batches(lo, hi, Bsub)
    orders = [torch.from_numpy(batch_order(SEED_ORDER, j, L)).to(dev) for a, b, j in subs]
    uniforms = [torch.from_numpy(sequence_uniforms(SEED_UNIFORM, a, b, (L,))).to(dev) for a, b, j in subs]
    local = torch.empty((hi - lo, L), dtype=torch.long, device=dev)

    def one_step():
        local.fill_(tokenizer_mask)
        for (a, b, j), order, uni in zip(subs, orders, uniforms):
            if job.dry:
                standin_oadm_sample(local[a - lo:b - lo], order, uni, n_draw)
            else:
                model.oadm_sample(local[a - lo:b - lo], order, uni, n_draw)
                if rank == 0 and len(subs) > 2:        # a long job keeps its log moving (stderr; stdout stays one line)
                    torch.cuda.synchronize(dev)
                    print(f"[bench] sub-batch {a}..{b} of shard {lo}..{hi} done", file=sys.stderr, flush=True)
        return gather_sequences(local, N)                       # RCCL all-gather when world > 1

    if model is not None:
        model.hip_handle(dev)
    for _ in range(args.warmup):
        one_step()
    job.fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full = one_step()
    job.fence()
    elapsed = job.max_over_ranks(time.perf_counter() - t0)
    assert full.shape == (N, L) and int((full == tokenizer_mask).sum()) == 0 and int(full.max()) < n_draw
    if rank == 0:
        rate = N * args.steps / elapsed
        out = {
            "metric": f"sequences/sec, OA_DM_{args.model} len-{L} sampling", "value": rate, "unit": "sequences/sec",
            "n_gpus": job.group_size(), "rccl_ranks": job.group_size(), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.precision,
            "data": "synthetic",
            "config": {"workload": f"OA_DM_{args.model} unconditional generate_oaardm, {N} sequences sharded over "
                                   f"{world} GPU(s), sub-batches of {Bsub}, seq_len {L}"
                                   + (" [DRY RUN: CPU stand-in sampler, not a measurement]" if job.dry else ""),
                       "global_batch": N, "seq_len": L, "parallelism": f"dp{world} (sequence sharding)"},
            "tokens_sha256": tokens_digest(full),
        }
        if model is not None:
            peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_FP32_TFLOPS
            f_eff = f_eff_per_sequence(model.hparams, L) * L
            out["step_roofline"] = {"flops_per_sequence": f_eff, "achieved": f_eff * rate / 1e12, "unit": "TFLOP/s",
                                    "frac": f_eff * rate / 1e12 / (peak * world)}
        print(json.dumps(out), flush=True)
    job.close()


def executed_flops_per_batch(model, sample0, order, uni, n_draw):
    """FLOPs the library actually launches for one batch of OA-DM sampling (dependency-cone pruning and the incremental
    leading layers skip part of the algorithmic count): the probe sums the in-range-tap FLOPs of every launch of an
    eagerly run step; step 0 of a call evaluates every row, the others only what changed.  Untimed."""
    def probed(n):
        model.probe(1)
        model.probe_read()
        model.oadm_sample(sample0.clone(), order[:n], uni[:n], n_draw)
        return sum(v["flops"] for v in model.probe_read().values())
    first = probed(1)
    rest = probed(2) - first
    model.probe(0)
    return first + (len(order) - 1) * rest


def self_launch(args):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a child job (this process never touches the
    GPU) and pass its exit status on."""
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seq-len", type=int, default=256)
    ap.add_argument("--model", default="640M", choices=["640M", "38M"])
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--sampler", default="oadm", choices=["oadm", "d3pm"],
                    help="oadm: BASELINE config 2 (the metric's configuration, default); d3pm: config 3 "
                         "(D3PM_BLOSUM, every Q_t transition step of generate_d3pm per batch)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-spot-check", action="store_true",
                    help="skip the check of the last timed batch against the CPU oracle (after the timed region)")
    ap.add_argument("--probe-every", type=int, default=64)
    ap.add_argument("--total-sequences", type=int, default=0,
                    help="strong-scaling form (BASELINE config 4): a step samples this many sequences in total, sharded "
                         "over the ranks (evodiff_amd.parallel.shard_range) and sampled in sub-batches of --batch")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU stand-in sampler, gloo: exercises launching, sharding, seeding and gathering only")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))

    job = Job(args)
    dev, world, rank = job.dev, job.world, job.rank
    d3pm = args.sampler == "d3pm"
    B, L = args.batch, args.seq_len
    model = None
    if job.dry:
        if d3pm:
            raise SystemExit("--dry-run covers the OA-DM forms")
        mask_id, n_draw = 28, 21
    else:
        from evodiff_amd import pretrained
        torch.manual_seed(0)                      # same synthetic weights on every rank
        if d3pm:
            loader = pretrained.D3PM_BLOSUM_640M if args.model == "640M" else pretrained.D3PM_BLOSUM_38M
            with contextlib.redirect_stdout(sys.stderr):   # the schedule builder prints its name, as the reference's does
                model, _, tokenizer, _, T, Q_bar, Q = loader(return_all=True, random_init=True)
            Q, Q_bar = Q.to(dev), Q_bar.to(dev)
        else:
            loader = pretrained.OA_DM_640M if args.model == "640M" else pretrained.OA_DM_38M
            model, _, tokenizer, _ = loader(random_init=True)
            sharpen_decoder(model, DECODER_SCALE)
        model.precision = args.precision
        model.use_graph = not args.no_graph
        mask_id, n_draw = tokenizer.mask_id, len(tokenizer.all_aas) - 6

    if args.total_sequences > 0:
        if d3pm:
            raise SystemExit("--total-sequences is the OA-DM form of config 4")
        return strong_scaling(args, job, model, mask_id, n_draw)

    from evodiff_amd.parallel import batch_order, sequence_uniforms
    total = args.warmup + args.steps
    # independent sequences: at step i rank r samples global batch j = i * world + r, i.e. the sequences
    # [j * B, (j + 1) * B) of the job, with that batch's order and those sequences' uniform streams
    batches = [i * world + rank for i in range(total)]
    if d3pm:                                   # generate.py:266: a uniformly random start over the K classes
        uniforms = [torch.from_numpy(np.moveaxis(sequence_uniforms(SEED_UNIFORM, j * B, (j + 1) * B, (T, L)), 2, 1)
                                     .copy()).to(dev) for j in batches]
        samples = [torch.from_numpy(np.random.default_rng([SEED_START, j]).integers(0, tokenizer.K, (B, L))).to(dev)
                   for j in batches]
    else:
        orders = [torch.from_numpy(batch_order(SEED_ORDER, j, L)).to(dev) for j in batches]
        uniforms = [torch.from_numpy(sequence_uniforms(SEED_UNIFORM, j * B, (j + 1) * B, (L,))).to(dev) for j in batches]
        samples = [torch.full((B, L), mask_id, dtype=torch.long, device=dev) for _ in range(total)]
    gathered = [torch.empty((world * B, L), dtype=torch.long, device=dev) for _ in range(total)] if job.collective else None

    def one_step(i):
        if job.dry:
            standin_oadm_sample(samples[i], orders[i], uniforms[i], n_draw)
        elif d3pm:
            model.d3pm_sample(samples[i], Q, Q_bar, uniforms[i], tokenizer.K)      # all T - 1 reverse steps
        else:
            model.oadm_sample(samples[i], orders[i], uniforms[i], n_draw)
        if job.collective:
            import torch.distributed as dist
            dist.all_gather_into_tensor(gathered[i], samples[i])   # RCCL over xGMI: the finished sequences

    if model is not None:
        model.hip_handle(dev)                  # weight packing / upload: outside the timed region
    for i in range(args.warmup):
        one_step(i)                            # includes graph capture
    if model is not None:
        model.probe(args.probe_every)
        if args.warmup:
            model.probe_read()
    job.fence()
    t0 = time.perf_counter()
    for i in range(args.warmup, total):
        one_step(i)
    job.fence()
    elapsed = job.max_over_ranks(time.perf_counter() - t0)
    stats = None
    if model is not None:
        stats = model.probe_read()
        model.probe(0)
    # every timed step's sequences over all ranks = global batches warmup*world .. total*world - 1, in order
    done = torch.cat((gathered if job.collective else samples)[args.warmup:], dim=0)
    if d3pm:
        assert int(done.min()) >= 0 and int(done.max()) < tokenizer.K - 6, "last D3PM step draws canonical residues"
    else:
        assert int((done == mask_id).sum()) == 0 and int(done.max()) < n_draw, "sampling left masks behind"

    if rank == 0:
        n_fwd = (T - 1) if d3pm else L                          # denoiser forwards per sequence
        name = f"{'D3PM_BLOSUM' if d3pm else 'OA_DM'}_{args.model}"
        out = {
            "metric": f"sequences/sec, {name} len-{L} sampling",
            "value": world * B * args.steps / elapsed,
            "unit": "sequences/sec",
            "n_gpus": job.group_size(), "rccl_ranks": job.group_size(), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic" + ("" if d3pm or job.dry else f" (seed-0 default init, decoder weight x {DECODER_SCALE:g})"),
            "config": {"workload": f"{name} unconditional {'generate_d3pm' if d3pm else 'generate_oaardm'}, "
                                   f"batch {B} per GPU, seq_len {L} ({n_fwd} diffusion steps per batch), hipGraph replay "
                                   f"{'off' if args.no_graph else 'on'}"
                                   + (" [DRY RUN: CPU stand-in sampler, not a measurement]" if job.dry else ""),
                       "global_batch": world * B, "seq_len": L, "parallelism": f"dp{world} (sequence sharding)"},
            # the timed steps' sequences in global order: the same global batches give the same digest at any N
            # (e.g. --gpus 1 --steps 2 --warmup 0 and --gpus 2 --steps 1 --warmup 0 both sample batches 0 and 1)
            "tokens_sha256": tokens_digest(done),
        }
        if model is not None:
            conv = stats["conv"]
            peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_FP32_TFLOPS
            achieved = conv["flops"] / (conv["ms"] * 1e-3) / 1e12 if conv["ms"] > 0 else 0.0
            f_eff = f_eff_per_sequence(model.hparams, L) * n_fwd
            seq_rate = world * B * args.steps / elapsed
            out["roofline"] = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                               "frac": achieved / peak, "traffic": measured_traffic(args),
                               "kernel": "conv_window_kernel (dilated k=5 convolution, activation window resident in LDS; since "
                                         "round 2 it also applies the LayerNorm + GELU behind it in its epilogue, and the "
                                         "LayerNorm launch that followed it in round 1, 15.6 us, is gone)",
                               "launches_timed": conv["launches"],
                               "avg_launch_ms": conv["ms"] / max(conv["launches"], 1),
                               "flops_per_launch": conv["flops"] / max(conv["launches"], 1)}
            out["kernel_ms_share"] = {k: v["ms"] for k, v in stats.items()}
            # whole step against the same peak.  `frac`: ALGORITHMIC FLOPs of the model on this batch (SURVEY.md §8d:
            # every position of every block, in-range taps only) / wall time - the figure §8d defines, comparable
            # across rounds.  `frac_executed`: the FLOPs the library really launched / wall time - hardware utilisation.
            step = {"flops_per_sequence": f_eff, "achieved": f_eff * seq_rate / 1e12, "unit": "TFLOP/s",
                    "frac": f_eff * seq_rate / 1e12 / (peak * world)}
            if args.precision == "bf16":
                out["roofline"]["frac_of_sustained_mfma"] = achieved / SUSTAINED_BF16_TFLOPS
                step["frac_of_sustained_mfma"] = step["achieved"] / (SUSTAINED_BF16_TFLOPS * world)
                step["sustained_mfma_note"] = ("a bare bf16 MFMA loop on random operands sustains 1953 TFLOP/s on this chip "
                                               "(78 % of the nominal 2500; profiles/r02_mfma_sustained_peak_microbench.txt)")
            if not d3pm:
                exe = executed_flops_per_batch(model, torch.full((B, L), mask_id, dtype=torch.long, device=dev),
                                               orders[-1], uniforms[-1], n_draw) / B
                step.update({"executed_flops_per_sequence": exe, "achieved_executed": exe * seq_rate / 1e12,
                             "frac_executed": exe * seq_rate / 1e12 / (peak * world),
                             "note": "frac counts every position of every block (algorithmic); dependency-cone pruning "
                                     "and the incremental leading layers skip the difference to frac_executed"})
            else:
                step["note"] = "every position of every block is evaluated at every D3PM step: executed = algorithmic"
            out["step_roofline"] = step
            if B * L <= 4096:
                # weight-streaming regime (north_star: small-batch steps against the 8 TB/s HBM peak): every diffusion
                # step reads each bf16 (fp32) weight once; bytes = parameters x element size, per step, over wall time
                n_param = sum(p.numel() for p in model.parameters())
                wbytes = n_param * (2 if args.precision == "bf16" else 4)
                gbs = wbytes * n_fwd * args.steps / elapsed / 1e9
                out["small_batch_roofline"] = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                               "frac": gbs / PEAK_HBM_GBS, "weight_bytes_per_step": wbytes}
            if world == 1 and not args.no_cpu_baseline and not d3pm:  # the CPU leg times the metric's OA-DM loop
                out["cpu_baseline"] = cpu_baseline(L)
            if not d3pm and not args.no_spot_check:
                # correctness evidence of THIS run, outside the timed region: rank 0's last timed batch against the oracle
                out["oracle_spot_check"] = oracle_spot_check(model, args.model, samples[-1], orders[-1], uniforms[-1].cpu(),
                                                             args.precision)
        print(json.dumps(out), flush=True)
    job.close()


if __name__ == "__main__":
    main()
