"""Random-geometry parity sweep on the GPU (a tool, not part of the suite: the suite holds the fixed cases that came out
of it).  For each seed: a random model geometry (width, slim or not, depth, causal or not, with or without timesteps),
a random batch / length, then
  - forward in fp32 and bf16 against the CPU oracle (tests/test_gpu_parity.py's tolerances),
  - non-causal without timesteps: a free-running OA-DM loop in fp32 with default options, checked draw by draw against
    the oracle teacher-forced along the GPU's trajectory, and bit-equal against the loop with prune / incremental / graph off,
  - causal: a cached autoregressive loop in bf16 (weight-streaming kernels for B <= 8) checked per step against the
    oracle's causal forward of the sampled sequences.
Usage: python tools/fuzz_parity.py [first_seed [count]]; prints one line per case and exits non-zero on the first failure."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from oracle import bytenet_ref as R  # noqa: E402
from test_gpu_parity import BF16_TOL, DRAW_MARGIN_FP32, FP32_TOL, product_model, rel_err  # noqa: E402

dev = torch.device("cuda", 0)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 24


def case(seed):
    rng = np.random.default_rng(seed)
    d_model = int(rng.choice([64, 128, 192, 256, 320, 384, 512, 640, 768, 1024, 1280]))
    cfg = dict(d_embed=8, d_model=d_model, n_layers=int(rng.integers(1, 11 if d_model <= 640 else 6)), kernel_size=5,
               r=int(rng.choice([1, 4, 16, 128])), slim=bool(rng.integers(0, 2)), activation=str(rng.choice(["gelu", "relu"])))
    causal = bool(rng.integers(0, 3) == 0)
    timesteps = None if causal or rng.integers(0, 2) else 500
    B = int(rng.integers(1, 13))
    L = int(rng.choice([1, 2, 5, 31, 64, 100, 127, 128, 129, 255, 256, 257, 300, 511, 600])) if d_model <= 640 else \
        int(rng.choice([3, 64, 130, 256]))
    if causal:
        cfg.update(causal=True, n_tokens=30)
    tag = f"seed {seed}: {cfg} timesteps={timesteps} B={B} L={L}"
    oracle = R.build_model(cfg, timesteps, 100 + seed)
    with torch.no_grad():
        oracle.decoder.conv.weight.mul_(4.0)
    m = product_model(oracle, cfg, timesteps=timesteps)
    n_tok = cfg.get("n_tokens", 31)
    x = torch.from_numpy(rng.integers(0, n_tok, (B, L)))
    y = torch.from_numpy(rng.integers(0, timesteps, (B,))) if timesteps else torch.zeros(B, dtype=torch.long)
    with torch.no_grad():
        ref = oracle(x, y).numpy()
    e32 = rel_err(m(x.to(dev), y.to(dev)).cpu().numpy(), ref)
    m.precision = "bf16"
    e16 = rel_err(m(x.to(dev), y.to(dev)).cpu().numpy(), ref)
    assert e32 < FP32_TOL and e16 < BF16_TOL, f"{tag}: forward fp32 {e32:.2e} bf16 {e16:.2e}"
    extra = ""
    if not causal and timesteps is None and L >= 2:
        m.precision = "fp32"
        n = min(L, 24)
        order = torch.from_numpy(rng.permutation(L)[:n].astype(np.int32))
        uni = torch.from_numpy(rng.random((n, B), dtype=np.float32))

        def run(**opts):
            for k, v in opts.items():
                if k == "graph":
                    m.use_graph = bool(v)
                else:
                    m.set_option(k, v)
            s = torch.full((B, L), R.MASK_ID, dtype=torch.long, device=dev)
            p = m.oadm_sample(s, order.to(dev), uni.to(dev), R.OADM_DRAW, return_probs=True)
            return s.cpu(), p.cpu()

        s0, p0 = run(prune=1, incremental=1, graph=1)
        s1, p1 = run(prune=0, incremental=0, graph=0)
        assert torch.equal(s0, s1) and torch.equal(p0, p1), f"{tag}: options change the fp32 loop"
        m.set_option("prune", 1)
        m.set_option("incremental", 1)
        m.use_graph = True
        safe = 0
        for s in range(n):
            state = s0.clone()
            state[:, order[s:].long()] = R.MASK_ID
            with torch.no_grad():
                lg = oracle(state, torch.zeros(B, dtype=torch.long))[:, int(order[s]), :R.OADM_DRAW]
            pr = torch.softmax(lg, dim=-1)
            assert float((pr - p0[s]).abs().max()) < 1e-4, f"{tag}: step {s} probabilities"
            idx, margin = R.inverse_cdf(pr, uni[s])
            ok = margin > DRAW_MARGIN_FP32
            assert torch.equal(idx[ok], s0[:, int(order[s])][ok]), f"{tag}: step {s} draws"
            safe += int(ok.sum())
        extra = f", OA-DM {n} steps ok ({safe} draws compared)"
    if causal and L >= 2:
        m.precision = "bf16"
        n = min(L - 1, 40)
        uni = torch.from_numpy(rng.random((n, B), dtype=np.float32))
        s = torch.full((B, n + 1), 29, dtype=torch.long, device=dev)
        p = m.ar_sample(s, uni.to(dev), n_steps=n, return_probs=True).cpu()
        s = s.cpu()
        with torch.no_grad():
            pr = torch.softmax(oracle(s, torch.zeros(B, dtype=torch.long)), dim=-1)[:, :n]
        d = float((p.permute(1, 0, 2) - pr).abs().max())
        assert d < 5e-2, f"{tag}: autoregressive probabilities differ by {d:.3f}"
        for b in range(B):
            idx, mg = R.inverse_cdf(p[:, b], uni[:, b])
            ok = mg > 1e-6
            assert torch.equal(idx[ok], s[b, 1:][ok]), f"{tag}: autoregressive draws"
        extra = f", AR {n} steps ok (max |dp| {d:.1e})"
    m.check_errors()
    print(f"{tag}: fp32 {e32:.1e} bf16 {e16:.1e}{extra}", flush=True)


t0 = time.time()
for seed in range(first, first + count):
    case(seed)
print(f"{count} cases in {time.time() - t0:.0f} s")
