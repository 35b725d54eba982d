#!/usr/bin/env python3
"""Throughput of the hot path on MI355X: OA_DM_640M unconditional sampling, length 256 (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full pass of the hot path over one batch: ``generate_oaardm``'s loop for B=64 sequences of
L=256 (256 denoiser forwards + draws) on every rank, followed by the all-gather of the finished sequences when
N > 1.  Weights are synthetic (reference-constructor default init under seed 0), inputs are resident in HBM
before the timed region.  Rank 0 prints ONE JSON line.

roofline: the dominant kernel is the windowed dilated convolution (conv_window_kernel: the convolution of every
block at L <= 256, 46 % of a step's time).  Its launches are timed live with HIP-event pairs on the launch stream on
every 64th diffusion step inside the timed region; achieved = algorithmic FLOPs of those launches (in-range taps
only) / their summed duration, against the 2.5 PFLOP/s dense bf16 MFMA peak.  traffic = memory-side
bytes of one such launch (dilation 1) from the committed PMC passes (profiles/, tools/pmc_traffic.sh), for the
default config only.  cpu_baseline: the CPU oracle (a port of the reference's PyTorch path) timed on this host for
a bounded sample.
"""
import argparse
import contextlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA
PEAK_FP32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E


def cpu_baseline(L, seconds_budget=20.0):
    """The CPU oracle on the same model and length, B=1, for a bounded number of diffusion steps."""
    from oracle import bytenet_ref as R
    threads = torch.get_num_threads()
    model = R.build_model(R.MODEL_CONFIGS["640M"], None, 0)
    rng = np.random.default_rng(0)
    order = rng.permutation(L)
    uni = torch.from_numpy(rng.random((L, 1), dtype=np.float32))
    sample = torch.full((1, L), R.MASK_ID, dtype=torch.long)
    sample, *_ = R.oadm_step(model, sample, int(order[0]), uni[0])          # warm-up
    t0 = time.perf_counter()
    n = 0
    while n < 8 or (time.perf_counter() - t0 < seconds_budget and n < L - 1):
        sample, *_ = R.oadm_step(model, sample, int(order[n + 1]), uni[n + 1])
        n += 1
    per_step = (time.perf_counter() - t0) / n
    return {"value": 1.0 / (per_step * L), "unit": "sequences/sec", "cores": threads, "kind": "port",
            "sample": f"oracle (fp32 PyTorch CPU) OA_DM_640M B=1 L={L}: {n} diffusion steps timed, "
                      f"{per_step:.3f} s/step, extrapolated to {L} steps per sequence"}


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
    path = os.path.join(ROOT, "profiles", "r01_conv_window_traffic.json")
    if not (args.model == "640M" and args.batch == 64 and args.seq_len == 256 and args.precision == "bf16"):
        return None
    try:
        with open(path) as f:
            return json.load(f)["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def strong_scaling(args, model, tokenizer, dev, world, rank):
    """BASELINE config 4's form: a fixed number of sequences, sharded over the ranks (contiguous shards, no collective
    on the data path), each rank sampling its shard in sub-batches; the finished sequences are all-gathered once per
    step.  One "step" = the whole job.  Order and uniforms of a sub-batch are seeded by the global index of its first
    sequence, so the sampled sequences do not depend on the number of ranks when the shards split on sub-batch
    boundaries."""
    import torch.distributed as dist
    from evodiff_amd.parallel import gather_sequences, shard_range
    N, L, Bsub = args.total_sequences, args.seq_len, args.batch
    n_draw = len(tokenizer.all_aas) - 6
    lo, hi = shard_range(N, rank, world)
    subs = [(a, min(a + Bsub, hi)) for a in range(lo, hi, Bsub)]
    orders, uniforms = [], []
    for a, b in subs:
        orders.append(torch.from_numpy(np.random.default_rng(1000 + a).permutation(L).astype(np.int32)).to(dev))
        gen = torch.Generator(device=dev)
        gen.manual_seed(2000 + a)
        uniforms.append(torch.rand((L, b - a), generator=gen, device=dev, dtype=torch.float32))
    local = torch.empty((hi - lo, L), dtype=torch.long, device=dev)

    def one_step():
        local.fill_(tokenizer.mask_id)
        for (a, b), order, uni in zip(subs, orders, uniforms):
            model.oadm_sample(local[a - lo:b - lo], order, uni, n_draw)
            if rank == 0 and len(subs) > 2:            # a long job keeps its log moving (stderr; stdout stays one line)
                torch.cuda.synchronize(dev)
                print(f"[bench] sub-batch {a}..{b} of shard {lo}..{hi} done", file=sys.stderr, flush=True)
        return gather_sequences(local, N)                       # RCCL all-gather when world > 1

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    model.hip_handle(dev)
    for _ in range(args.warmup):
        one_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full = one_step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert full.shape == (N, L) and int((full == tokenizer.mask_id).sum()) == 0 and int(full.max()) < n_draw
    if rank == 0:
        peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_FP32_TFLOPS
        f_eff = f_eff_per_sequence(model.hparams, L) * L
        rate = N * args.steps / elapsed
        print(json.dumps({
            "metric": f"sequences/sec, OA_DM_{args.model} len-{L} sampling", "value": rate, "unit": "sequences/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.precision,
            "data": "synthetic",
            "config": {"workload": f"OA_DM_{args.model} unconditional generate_oaardm, {N} sequences sharded over "
                                   f"{world} GPU(s), sub-batches of {Bsub}, seq_len {L}",
                       "global_batch": N, "seq_len": L, "parallelism": f"dp{world} (sequence sharding)"},
            "step_roofline": {"flops_per_sequence": f_eff, "achieved": f_eff * rate / 1e12, "unit": "TFLOP/s",
                              "frac": f_eff * rate / 1e12 / (peak * world)},
        }), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


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
    ap.add_argument("--probe-every", type=int, default=64)
    ap.add_argument("--total-sequences", type=int, default=0,
                    help="strong-scaling form (BASELINE config 4): a step samples this many sequences in total, sharded "
                         "over the ranks (evodiff_amd.parallel.shard_range) and sampled in sub-batches of --batch")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from evodiff_amd import pretrained
    torch.manual_seed(0)                      # same synthetic weights on every rank
    d3pm = args.sampler == "d3pm"
    if d3pm:
        loader = pretrained.D3PM_BLOSUM_640M if args.model == "640M" else pretrained.D3PM_BLOSUM_38M
        with contextlib.redirect_stdout(sys.stderr):       # the schedule builder prints its name, as the reference's does
            model, _, tokenizer, _, T, Q_bar, Q = loader(return_all=True, random_init=True)
        Q, Q_bar = Q.to(dev), Q_bar.to(dev)
    else:
        loader = pretrained.OA_DM_640M if args.model == "640M" else pretrained.OA_DM_38M
        model, _, tokenizer, _ = loader(random_init=True)
    model.precision = args.precision
    model.use_graph = not args.no_graph
    B, L = args.batch, args.seq_len
    n_draw = len(tokenizer.all_aas) - 6

    if args.total_sequences > 0:
        if d3pm:
            raise SystemExit("--total-sequences is the OA-DM form of config 4")
        return strong_scaling(args, model, tokenizer, dev, world, rank)

    total = args.warmup + args.steps
    # independent sequences: each rank samples its own shard with its own order/uniform streams
    rng = np.random.default_rng(1000 + rank)
    gen = torch.Generator(device=dev)
    gen.manual_seed(2000 + rank)
    if d3pm:                                   # generate.py:266: a uniformly random start over the K classes
        uniforms = [torch.rand((T, B, L), generator=gen, device=dev, dtype=torch.float32) for _ in range(total)]
        samples = [torch.from_numpy(rng.integers(0, tokenizer.K, (B, L))).to(dev) for _ in range(total)]
    else:
        orders = [torch.from_numpy(rng.permutation(L).astype(np.int32)).to(dev) for _ in range(total)]
        uniforms = [torch.rand((L, B), generator=gen, device=dev, dtype=torch.float32) for _ in range(total)]
        samples = [torch.full((B, L), tokenizer.mask_id, dtype=torch.long, device=dev) for _ in range(total)]
    gathered = torch.empty((world * B, L), dtype=torch.long, device=dev) if world > 1 else None

    def one_step(i):
        if d3pm:
            model.d3pm_sample(samples[i], Q, Q_bar, uniforms[i], tokenizer.K)      # all T - 1 reverse steps
        else:
            model.oadm_sample(samples[i], orders[i], uniforms[i], n_draw)
        if world > 1:
            dist.all_gather_into_tensor(gathered, samples[i])      # RCCL over xGMI: the finished sequences

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    model.hip_handle(dev)                      # weight packing / upload: outside the timed region
    for i in range(args.warmup):
        one_step(i)                            # includes graph capture
    model.probe(args.probe_every)
    if args.warmup:
        model.probe_read()
    fence()
    t0 = time.perf_counter()
    for i in range(args.warmup, total):
        one_step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stats = model.probe_read()
    model.probe(0)
    done = samples[-1]
    if d3pm:
        assert int(done.min()) >= 0 and int(done.max()) < tokenizer.K - 6, "last D3PM step draws canonical residues"
    else:
        assert int((done == tokenizer.mask_id).sum()) == 0 and int(done.max()) < n_draw, "sampling left masks behind"

    if rank == 0:
        conv = stats["conv"]
        peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_FP32_TFLOPS
        achieved = conv["flops"] / (conv["ms"] * 1e-3) / 1e12 if conv["ms"] > 0 else 0.0
        n_fwd = T - 1 if d3pm else L                           # denoiser forwards per sequence
        f_eff = f_eff_per_sequence(model.hparams, L) * n_fwd
        name = f"{'D3PM_BLOSUM' if d3pm else 'OA_DM'}_{args.model}"
        out = {
            "metric": f"sequences/sec, {name} len-{L} sampling",
            "value": world * B * args.steps / elapsed,
            "unit": "sequences/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"{name} unconditional {'generate_d3pm' if d3pm else 'generate_oaardm'}, "
                                   f"batch {B} per GPU, seq_len {L} ({n_fwd} diffusion steps per batch), hipGraph replay "
                                   f"{'off' if args.no_graph else 'on'}",
                       "global_batch": world * B, "seq_len": L, "parallelism": f"dp{world} (sequence sharding)"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": measured_traffic(args),
                         "kernel": "conv_window_kernel (dilated k=5 convolution, activation window resident in LDS)",
                         "launches_timed": conv["launches"],
                         "avg_launch_ms": conv["ms"] / max(conv["launches"], 1),
                         "flops_per_launch": conv["flops"] / max(conv["launches"], 1)},
            "kernel_ms_share": {k: v["ms"] for k, v in stats.items()},
            # whole step against the same peak: algorithmic FLOPs of the model on this batch (in-range
            # taps only; work the pruning skips is NOT subtracted) / wall time
            "step_roofline": {"flops_per_sequence": f_eff, "achieved": f_eff * world * B * args.steps / elapsed / 1e12,
                              "unit": "TFLOP/s", "frac": f_eff * world * B * args.steps / elapsed / 1e12 / (peak * world),
                              "note": "denominator counts every position of every block" + ("" if d3pm else
                                      "; dependency-cone pruning and the incremental leading layers skip ~15 % of it "
                                      "in the OA-DM path")},
        }
        if B * L <= 4096:
            # weight-streaming regime (north_star: small-batch steps against the 8 TB/s HBM peak): every diffusion step
            # reads each bf16 (fp32) weight once; bytes = parameters x element size, per step, over the wall time
            n_param = sum(p.numel() for p in model.parameters())
            wbytes = n_param * (2 if args.precision == "bf16" else 4)
            gbs = wbytes * n_fwd * args.steps / elapsed / 1e9
            out["small_batch_roofline"] = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                           "frac": gbs / PEAK_HBM_GBS, "weight_bytes_per_step": wbytes}
        if world == 1 and not args.no_cpu_baseline and not d3pm:      # the CPU leg times the metric's OA-DM loop
            out["cpu_baseline"] = cpu_baseline(L)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
