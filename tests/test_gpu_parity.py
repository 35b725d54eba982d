"""GPU parity: libevodiff_hip.so (through the C ABI, via evodiff_amd) against the CPU oracle and the golden
fixtures.  Run on an MI355X with ``pytest -m gpu``.

Tolerances (north_star): fp32 validation path — per-step logits within 1e-4 relative (max |Δ| over the
tensor divided by max |ref|); bf16 fast path — within BF16_TOL stated below; sampled tokens bit-exact
wherever the oracle's draw is not within DRAW_MARGIN of a category boundary.
"""
import os

import numpy as np
import pytest
import torch

from oracle import bytenet_ref as R

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-4
# relative to max |logit|; bf16 operands (8-bit mantissa) through up to 56 residual blocks.  Measured against the oracle on
# these tests' shapes (tools/bf16_error_survey.py, profiles/r02_bf16_error_survey.txt): 6.3e-3 ... 1.3e-2 (largest:
# 640M, L = 512); the stated tolerance is 1.5 x the largest measured value (round 1 stated 3e-2 without the survey).
BF16_TOL = 2e-2
DRAW_MARGIN_FP32 = 1e-4  # probability mass; draws closer than this to a CDF boundary may differ
DRAW_MARGIN_BF16 = 5e-2

TINY = dict(d_embed=8, d_model=64, n_layers=8, kernel_size=5, r=128, slim=True, activation="gelu")
TINY_WIDE = dict(d_embed=8, d_model=256, n_layers=9, kernel_size=5, r=128, slim=False, activation="gelu")
MID = dict(d_embed=8, d_model=384, n_layers=10, kernel_size=5, r=128, slim=True, activation="gelu")
# shapes the tuned bf16 kernel takes: N in {256, 512} (BN=256) and N = 640 (BN=320)
BIG512 = dict(d_embed=8, d_model=512, n_layers=9, kernel_size=5, r=128, slim=True, activation="gelu")
BIG640 = dict(d_embed=8, d_model=640, n_layers=8, kernel_size=5, r=128, slim=False, activation="gelu")
WIDE1280 = dict(d_embed=8, d_model=1280, n_layers=10, kernel_size=5, r=128, slim=False, activation="gelu")  # 640M's block


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda", 0)


def product_model(oracle_model, cfg, timesteps=None, precision="fp32"):
    from evodiff_amd.model import ByteNetLMTime
    m = ByteNetLMTime(cfg.get("n_tokens", len(R.MSA_ALPHABET)), cfg["d_embed"], cfg["d_model"], cfg["n_layers"],
                      cfg["kernel_size"], cfg["r"], causal=cfg.get("causal", False), padding_idx=R.MASK_ID, rank=None,
                      dropout=0.0, tie_weights=False,
                      final_ln=True, slim=cfg["slim"], activation=cfg["activation"], timesteps=timesteps)
    m.load_state_dict(oracle_model.state_dict())
    m.precision = precision
    return m.eval()


def _sd_digest(sd):
    """SHA-256 over keys and fp32 bytes of a state dict, as oracle/make_golden.py stores it (``sd_sha``)."""
    import hashlib
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(np.ascontiguousarray(v.detach().numpy()).tobytes())
    return h.hexdigest()


def rel_err(got, ref):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30))


def test_library_loaded_is_the_in_tree_hip_build():
    from evodiff_amd import _lib
    lib = _lib.load()
    assert lib.ed_abi_version() == 1
    assert os.path.samefile(_lib.LIB_PATH, os.path.join(os.path.dirname(os.path.dirname(__file__)), "evodiff_amd",
                                                        "csrc", "libevodiff_hip.so"))


def test_tiny_forward_against_reference_golden(golden_dir, dev):
    """Weights, inputs and outputs all come from the reference's own model.py (fixture)."""
    g = np.load(os.path.join(golden_dir, "tiny_forward.npz"))
    oracle = R.build_model(TINY, None, 11)
    oracle.load_state_dict({k: torch.from_numpy(g["sd/" + k]) for k in oracle.state_dict().keys()})
    m = product_model(oracle, TINY, precision="fp32")
    for L in (7, 100, 300):
        x = torch.from_numpy(g[f"x_L{L}"]).to(dev)
        logits, acts = m(x, torch.zeros(2, dtype=torch.long, device=dev), return_blocks=True)
        assert rel_err(logits.cpu().numpy(), g[f"logits_L{L}"]) < FP32_TOL
        if g[f"blocks_L{L}"].size:
            ref = g[f"blocks_L{L}"]
            for i in range(ref.shape[0]):
                assert rel_err(acts[i].cpu().numpy(), ref[i]) < FP32_TOL, f"block {i}"
    m.precision = "bf16"
    x = torch.from_numpy(g["x_L100"]).to(dev)
    assert rel_err(m(x, torch.zeros(2, dtype=torch.long, device=dev)).cpu().numpy(), g["logits_L100"]) < BF16_TOL


def test_time_encoding_quirk_against_reference_golden(golden_dir, dev):
    g = np.load(os.path.join(golden_dir, "tiny_timeenc.npz"))
    oracle = R.build_model(TINY, 500, 13)
    m = product_model(oracle, TINY, timesteps=500, precision="fp32")
    x, y = torch.from_numpy(g["x"]).to(dev), torch.from_numpy(g["y"]).to(dev)
    logits, acts = m(x, y, return_blocks=True)
    assert rel_err(acts[0].cpu().numpy(), g["embed"]) < 1e-6
    assert rel_err(logits.cpu().numpy(), g["logits"]) < FP32_TOL


@pytest.mark.parametrize("cfg,B,L", [(MID, 3, 1), (MID, 3, 37), (MID, 2, 128), (MID, 2, 300), (TINY_WIDE, 1, 513),
                                     (TINY_WIDE, 5, 256), (BIG512, 9, 300), (BIG512, 3, 256), (BIG640, 4, 512),
                                     (BIG640, 16, 130), (BIG640, 1, 1024), (BIG640, 5, 77)])
def test_forward_vs_oracle_shapes(dev, cfg, B, L):
    """Ragged and edge lengths: L=1, L below/at/above the 128-row tile, L > 2*max dilation."""
    oracle = R.build_model(cfg, None, 21)
    m = product_model(oracle, cfg)
    x = torch.from_numpy(np.random.default_rng(B * 1000 + L).integers(0, 31, (B, L)))
    with torch.no_grad():
        ref = oracle(x, torch.zeros(B, dtype=torch.long)).numpy()
    t = torch.zeros(B, dtype=torch.long, device=dev)
    assert rel_err(m(x.to(dev), t).cpu().numpy(), ref) < FP32_TOL
    m.precision = "bf16"
    assert rel_err(m(x.to(dev), t).cpu().numpy(), ref) < BF16_TOL


def test_38m_forward_vs_oracle_and_golden(golden_dir, dev):
    cfg = R.MODEL_CONFIGS["38M"]
    g = np.load(os.path.join(golden_dir, "m38_forward.npz"))
    oracle = R.build_model(cfg, None, 0)
    m = product_model(oracle, cfg)
    x = torch.from_numpy(g["x"])
    with torch.no_grad():
        ref = oracle(x, torch.zeros(2, dtype=torch.long)).numpy()
    t = torch.zeros(2, dtype=torch.long, device=dev)
    got = m(x.to(dev), t).cpu().numpy()
    assert rel_err(got, ref) < FP32_TOL
    # the fixture was produced by the reference constructor under the same seed: the seeded weights must be the very
    # tensors the reference built (SHA-256 over the state dict, stored in the fixture), and then the GPU logits must
    # match the reference's own output file - unconditionally
    assert _sd_digest(oracle.state_dict()) == str(g["sd_sha"]), \
        "seeded default init differs from the reference constructor's (another torch build?): regenerate the fixture"
    assert rel_err(got, g["logits"]) < FP32_TOL
    m.precision = "bf16"
    assert rel_err(m(x.to(dev), t).cpu().numpy(), ref) < BF16_TOL


def test_config1_38m_sampling_loop_b1_l100(dev):
    """BASELINE config 1, the reference notebook's call (examples/evodiff.ipynb cell 10: ``generate_oaardm(model,
    tokenizer, 100, batch_size=1)`` on OA_DM_38M; generate.py:180-215): B = 1, L = 100, all 100 steps, hipGraph replay
    on.  Teacher-forced against the oracle's trajectory in fp32 and bf16 (softmax row and token of every step), then
    the free-running loop through the Python API: fp32 reproduces the oracle's final tokens wherever no draw sat on a
    category boundary, bf16 stays a valid sample."""
    from evodiff_amd.generate import generate_oaardm
    from evodiff_amd.utils import Tokenizer
    cfg = R.MODEL_CONFIGS["38M"]
    oracle = R.build_model(cfg, None, 0)
    with torch.no_grad():
        oracle.decoder.conv.weight.mul_(8.0)       # peaky softmax rows: the draws are not all near-uniform
    B, L = 1, 100
    order = np.random.default_rng(100).permutation(L)
    uni = torch.from_numpy(np.random.default_rng(101).random((L, B), dtype=np.float32))
    trace = []
    final = R.generate_oaardm_ref(oracle, L, B, order, uni, trace=trace)
    margins = np.array([float(t["margin"].min()) for t in trace])
    for precision, ptol, mtol in (("fp32", 1e-5, DRAW_MARGIN_FP32), ("bf16", 5e-2, DRAW_MARGIN_BF16)):
        m = product_model(oracle, cfg, precision=precision)
        m.use_graph = True
        state = torch.full((B, L), R.MASK_ID, dtype=torch.long)
        n_safe = 0
        for s, tr in enumerate(trace):
            pos = int(order[s])
            sample = state.clone().to(dev)
            # one captured step per call; a second, traced call gives the softmax row (traced calls run eagerly)
            m.oadm_sample(sample, torch.tensor([pos]), uni[s:s + 1].to(dev), R.OADM_DRAW)
            again = state.clone().to(dev)
            probs = m.oadm_sample(again, torch.tensor([pos]), uni[s:s + 1].to(dev), R.OADM_DRAW, return_probs=True)
            assert torch.equal(sample, again), f"graph replay and eager launches differ at step {s}"
            assert np.abs(probs[0].cpu().numpy() - tr["p"].numpy()).max() < ptol, (precision, s)
            n_safe += _check_draws(sample[:, pos].cpu().numpy(), tr["sample"][:, pos].numpy(), tr["margin"].numpy(), mtol)
            state = tr["sample"]
        assert n_safe > 0.5 * L, "test is vacuous: most draws sit on a category boundary"
        # free-running, graph replay: all 100 steps in one call, as the notebook runs it
        sample, strings = generate_oaardm(m, Tokenizer(), L, batch_size=B, device=dev, order=order, uniforms=uni)
        assert int((sample == R.MASK_ID).sum()) == 0 and int(sample.max()) < R.OADM_DRAW and len(strings[0]) == L
        if precision == "fp32" and margins.min() > DRAW_MARGIN_FP32:
            assert torch.equal(sample.cpu(), final)
        elif precision == "fp32":
            first = int(np.argmax(margins <= DRAW_MARGIN_FP32))    # identical up to the first ambiguous draw
            assert torch.equal(sample.cpu()[:, order[:first]], final[:, order[:first]])


def _check_draws(got_tokens, ref_tokens, margins, margin_tol):
    got_tokens, ref_tokens, margins = map(np.asarray, (got_tokens, ref_tokens, margins))
    safe = margins > margin_tol
    assert np.array_equal(got_tokens[safe], ref_tokens[safe])
    return int(safe.sum())


@pytest.mark.parametrize("precision,tol,margin", [("fp32", FP32_TOL, DRAW_MARGIN_FP32), ("bf16", BF16_TOL, DRAW_MARGIN_BF16)])
def test_oadm_steps_teacher_forced(golden_dir, dev, precision, tol, margin):
    """Every step of the reference's generate_oaardm trace (fixture), restarted from the oracle's state."""
    g = np.load(os.path.join(golden_dir, "oadm_trace.npz"))
    oracle = R.build_model(TINY_WIDE, None, 12)
    with torch.no_grad():
        oracle.decoder.conv.weight.mul_(float(g["decoder_scale"]))
    m = product_model(oracle, TINY_WIDE, precision=precision)
    order, uni = g["order"], torch.from_numpy(g["uniforms"])
    B, L = 2, len(order)
    trace = []
    R.generate_oaardm_ref(oracle, L, B, order, uni, trace=trace)
    state = torch.full((B, L), R.MASK_ID, dtype=torch.long)
    n_safe = 0
    for s, tr in enumerate(trace):
        sample = state.clone().to(dev)
        probs = m.oadm_sample(sample, torch.tensor([order[s]]), uni[s:s + 1].to(dev), R.OADM_DRAW, return_probs=True)
        assert np.abs(probs[0].cpu().numpy() - g["p"][s]).max() < (1e-5 if precision == "fp32" else 5e-2)
        n_safe += _check_draws(sample[:, order[s]].cpu().numpy(), g["tok"][s], g["margin"][s], margin)
        others = np.ones(L, bool)
        others[order[s]] = False
        assert np.array_equal(sample.cpu().numpy()[:, others], state.numpy()[:, others])   # only column pos changes
        state = tr["sample"]
    assert n_safe > 0.3 * B * L, "test is vacuous: most draws sit on a category boundary"


def test_oadm_full_trajectory_fp32(golden_dir, dev):
    """Free-running L-step loop (hipGraph replay) reproduces the reference's final sequences in fp32."""
    from evodiff_amd.generate import generate_oaardm
    from evodiff_amd.utils import Tokenizer
    g = np.load(os.path.join(golden_dir, "oadm_trace.npz"))
    assert g["margin"].min() > DRAW_MARGIN_FP32, "fixture has a boundary-ambiguous draw; regenerate with another seed"
    oracle = R.build_model(TINY_WIDE, None, 12)
    with torch.no_grad():
        oracle.decoder.conv.weight.mul_(float(g["decoder_scale"]))
    m = product_model(oracle, TINY_WIDE, precision="fp32")
    for use_graph in (True, False):
        m.use_graph = use_graph
        sample, strings = generate_oaardm(m, Tokenizer(), len(g["order"]), batch_size=2, device=dev,
                                          order=g["order"], uniforms=g["uniforms"])
        assert np.array_equal(sample.cpu().numpy(), g["final"])
        assert strings == [str(s) for s in g["strings"]]


@pytest.mark.parametrize("tag", ["blosum", "uniform"])
def test_d3pm_against_reference_trace(golden_dir, dev, tag):
    from evodiff_amd.utils import DEFAULT_BLOSUM, Tokenizer
    g = np.load(os.path.join(golden_dir, f"d3pm_trace_{tag}.npz"))
    oracle = R.build_model(TINY, 500, 13)
    with torch.no_grad():
        oracle.decoder.conv.weight.mul_(float(g["decoder_scale"]))
    m = product_model(oracle, TINY, timesteps=500, precision="fp32")
    if tag == "blosum":
        tok = Tokenizer(path_to_blosum=DEFAULT_BLOSUM, sequences=True)
        Qbar, Q = tok.q_blosum_schedule(timesteps=500)
    else:
        tok = Tokenizer(sequences=True)
        Qbar, Q = tok.q_random_schedule(timesteps=500)
    uni = torch.from_numpy(g["uniforms"])
    trace = []
    R.generate_d3pm_ref(oracle, Q, Qbar, 500, torch.from_numpy(g["start"]), uni, K=tok.K, trace=trace)
    # teacher-forced single steps incl. t == 1 (restricted draw), with the posterior checked in fp64
    states = [torch.from_numpy(g["start"])] + [t["sample"] for t in trace]
    for s in (0, 1, 2, 250, 497, 498):
        t = 499 - s
        sample = states[s].clone().to(dev)
        post = m.d3pm_sample(sample, Q, Qbar, uni, tok.K, t_start=t, n_steps=1, return_post=True)
        ref_post = trace[s]["post"].numpy()
        assert np.abs(post[0].cpu().numpy() - ref_post).max() < 2e-5
        _check_draws(sample.cpu().numpy(), trace[s]["sample"].numpy(), trace[s]["margin"].numpy(), DRAW_MARGIN_FP32)
    # free-running 499 steps (hipGraph replay): chaotic, so only require agreement while no ambiguous draw occurred
    from evodiff_amd.generate import generate_d3pm
    sample, strings = generate_d3pm(m, tok, Q, Qbar, 500, 20, batch_size=2, device=dev, uniforms=uni,
                                    start=torch.from_numpy(g["start"]))
    if min(float(t["margin"].min()) for t in trace) > DRAW_MARGIN_FP32:
        assert np.array_equal(sample.cpu().numpy(), g["final"])
    assert sample.max() < tok.K - 6          # final step only emits canonical amino acids


def test_graph_replay_equals_eager_bf16(dev):
    oracle = R.build_model(MID, None, 5)
    m = product_model(oracle, MID, precision="bf16")
    B, L = 4, 96
    order = torch.from_numpy(np.random.default_rng(0).permutation(L).astype(np.int32))
    uni = torch.from_numpy(np.random.default_rng(1).random((L, B), dtype=np.float32)).to(dev)
    outs = []
    for use_graph in (True, False, True):
        m.use_graph = use_graph
        sample = torch.full((B, L), R.MASK_ID, dtype=torch.long, device=dev)
        m.oadm_sample(sample, order, uni, R.OADM_DRAW)
        outs.append(sample.cpu().numpy())
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    assert (outs[0] < R.OADM_DRAW).all()


@pytest.mark.parametrize("cfg,B,L", [(TINY, 3, 256), (TINY, 2, 100), (TINY_WIDE, 2, 96), (MID, 5, 64),
                                     (TINY_WIDE, 2, 37), (BIG640, 40, 256), (BIG512, 33, 128)])
def test_cone_pruning_changes_nothing(dev, cfg, B, L):
    """ED_OPT_PRUNE evaluates the trailing blocks only on the positions the sampled position depends on: the
    softmax rows and draws must equal those of the every-position evaluation (fp32: the same arithmetic per
    row, so bit-equal; bf16: the small problems take the other GEMM kernel, so only within tolerance)."""
    oracle = R.build_model(cfg, None, 21)
    n = min(L, 12)
    order = torch.from_numpy(np.random.default_rng(L).permutation(L)[:n].astype(np.int32))
    uni = torch.from_numpy(np.random.default_rng(1).random((n, B), dtype=np.float32)).to(dev)
    for precision in ("fp32", "bf16"):
        m = product_model(oracle, cfg, precision=precision)
        outs = []
        for prune, use_graph in ((0, False), (1, False), (1, True)):
            m.set_option("prune", prune)
            m.use_graph = use_graph
            sample = torch.full((B, L), R.MASK_ID, dtype=torch.long, device=dev)
            if use_graph:
                m.oadm_sample(sample, order, uni, R.OADM_DRAW)
                outs.append((sample.cpu().numpy(), None))
            else:
                probs = m.oadm_sample(sample, order, uni, R.OADM_DRAW, return_probs=True)
                outs.append((sample.cpu().numpy(), probs.cpu().numpy()))
        (tok0, p0), (tok1, p1), (tok2, _) = outs
        assert np.array_equal(tok1, tok2)                       # graph replay == eager, pruned
        if precision == "fp32":
            assert np.array_equal(p0, p1) and np.array_equal(tok0, tok1)
        else:
            # free-running: compare step 0 (identical inputs) by value, the rest only for being valid draws
            assert np.abs(p0[0] - p1[0]).max() < 2e-2
            assert (tok1[:, order.numpy()] < R.OADM_DRAW).all()


@pytest.mark.parametrize("cfg,B,L", [(TINY_WIDE, 3, 96), (MID, 5, 64), (TINY_WIDE, 2, 300), (BIG640, 9, 256),
                                     (BIG512, 33, 128)])
def test_incremental_layers_change_nothing(dev, cfg, B, L):
    """ED_OPT_INCREMENTAL recomputes, in the leading blocks, only the rows within the receptive field of the token
    the previous step filled.  Same arithmetic per row: in fp32 the softmax rows and tokens of a free-running loop are
    bit-equal to the every-row evaluation (eager and graph replay); in bf16 the window problems take the small-tile
    kernel, so the first steps are compared by value."""
    oracle = R.build_model(cfg, None, 31)
    n = min(L, 24)
    rng = np.random.default_rng(L + B)
    # positions near both ends and repeated neighbours stress the clamped windows
    order = np.concatenate([[0, 1, L - 1, L - 2, L // 2], rng.permutation(np.arange(2, L - 2))[: n - 5]]).astype(np.int32)
    order = torch.from_numpy(order)
    uni = torch.from_numpy(np.random.default_rng(1).random((n, B), dtype=np.float32)).to(dev)
    for precision in ("fp32", "bf16"):
        m = product_model(oracle, cfg, precision=precision)
        outs = []
        for inc, use_graph in ((0, False), (1, False), (1, True)):
            m.set_option("incremental", inc)
            m.use_graph = use_graph
            sample = torch.full((B, L), R.MASK_ID, dtype=torch.long, device=dev)
            if use_graph:
                m.oadm_sample(sample, order, uni, R.OADM_DRAW)
                outs.append((sample.cpu().numpy(), None))
            else:
                probs = m.oadm_sample(sample, order, uni, R.OADM_DRAW, return_probs=True)
                outs.append((sample.cpu().numpy(), probs.cpu().numpy()))
        (tok0, p0), (tok1, p1), (tok2, _) = outs
        assert np.array_equal(tok1, tok2)                       # graph replay == eager
        if precision == "fp32":
            assert np.array_equal(p0, p1) and np.array_equal(tok0, tok1)
        else:
            same = (tok0 == tok1).all(axis=1)                   # sequences whose draws agreed so far see the same inputs
            assert same.mean() > 0.5
            assert np.abs(p0[:3] - p1[:3]).max() < 3e-2
            assert np.abs(p0[-1][same] - p1[-1][same]).max() < 5e-2


@pytest.mark.parametrize("cfg,B,L", [(BIG640, 2, 100), (BIG512, 3, 256), (MID, 1, 37)])
def test_fused_split_k_layernorm_matches_separate_reduction(dev, cfg, B, L):
    """ED_OPT_FUSE_REDUCE_LN: the LayerNorm behind a split-K GEMM adds up the partial sums itself.  Same additions in
    the same order ahead of the statistics, so against the separate reduction launch only the order of the row sums
    differs: logits and softmax rows agree far inside the bf16 tolerance, and both stay within it of the oracle."""
    oracle = R.build_model(cfg, None, 41)
    m = product_model(oracle, cfg, precision="bf16")
    rng = np.random.default_rng(B * L)
    x = torch.from_numpy(rng.integers(0, 26, (B, L)))
    x[:, rng.permutation(L)[: L // 3]] = R.MASK_ID
    zeros = torch.zeros(B, dtype=torch.long)
    want = oracle(x, zeros).detach().numpy()
    n = min(L, 6)
    order = torch.from_numpy(rng.permutation(L)[:n].astype(np.int32))
    uni = torch.from_numpy(rng.random((n, B), dtype=np.float32)).to(dev)
    logits, probs = [], []
    for fuse in (1, 0):
        m.set_option("fuse_reduce_ln", fuse)
        logits.append(m(x.to(dev), zeros.to(dev)).cpu().numpy())
        m.use_graph = False
        sample = torch.full((B, L), R.MASK_ID, dtype=torch.long, device=dev)
        probs.append(m.oadm_sample(sample, order, uni, R.OADM_DRAW, return_probs=True).cpu().numpy())
        assert rel_err(logits[-1], want) < BF16_TOL
    assert rel_err(logits[0], logits[1]) < 4e-3
    assert np.abs(probs[0][0] - probs[1][0]).max() < 4e-3
    # 64 x 64 tiles for the same problems (ED_OPT_TILE64_MAX): another cut of the K loops, the same sums
    m.set_option("fuse_reduce_ln", 1)
    m.set_option("tile64_max", 100000)
    small = m(x.to(dev), zeros.to(dev)).cpu().numpy()
    assert rel_err(small, want) < BF16_TOL and rel_err(small, logits[0]) < 4e-3
    # the tuned kernel with a cut K loop for the same problems, where their shape allows it (ED_OPT_TUNED_SPLIT_MIN_ROWS)
    m.set_option("tuned_split_min_rows", 256)
    mid = m(x.to(dev), zeros.to(dev)).cpu().numpy()
    assert rel_err(mid, want) < BF16_TOL and rel_err(mid, logits[0]) < 1e-2
    m.set_option("tuned_split_min_rows", 0)
    m.precision = "fp32"                       # fp32 problems keep the 128 x 128 tiles: the validation path is unchanged
    assert rel_err(m(x.to(dev), zeros.to(dev)).cpu().numpy(), want) < FP32_TOL


def test_640m_steps_vs_oracle(dev):
    """OA_DM_640M geometry (config 2 at reduced batch): 8 teacher-forced steps, fp32 1e-4 and bf16."""
    cfg = R.MODEL_CONFIGS["640M"]
    oracle = R.build_model(cfg, None, 0)
    m = product_model(oracle, cfg, precision="fp32")
    B, L, n = 2, 256, 8
    rng = np.random.default_rng(0)
    order = rng.permutation(L)[:n]
    uni = torch.from_numpy(np.random.default_rng(1).random((n, B), dtype=np.float32))
    state = torch.full((B, L), R.MASK_ID, dtype=torch.long)
    state[:, rng.permutation(L)[: L // 2]] = torch.from_numpy(rng.integers(0, 20, (B, L // 2)))
    zeros = torch.zeros(B, dtype=torch.long, device=dev)
    for s in range(n):
        nxt, logits, p, margin = R.oadm_step(oracle, state, int(order[s]), uni[s])
        for precision, tol, mtol in (("fp32", FP32_TOL, DRAW_MARGIN_FP32), ("bf16", BF16_TOL, DRAW_MARGIN_BF16)):
            m.precision = precision
            if s in (0, n - 1):
                assert rel_err(m(state.to(dev), zeros).cpu().numpy(), logits.numpy()) < tol
            sample = state.clone().to(dev)
            m.oadm_sample(sample, torch.tensor([order[s]]), uni[s:s + 1].to(dev), R.OADM_DRAW)
            safe = margin.numpy() > mtol
            assert np.array_equal(sample[:, order[s]].cpu().numpy()[safe], nxt[:, order[s]].numpy()[safe])
        state = nxt


def test_full_size_batch_invariance_640m(dev):
    """BASELINE config 2 (B=64, L=256, 640M, bf16): sequences are independent, so rows of a 64-batch run
    must equal the same rows sampled in a batch of 2 (which test_640m_steps_vs_oracle ties to the oracle)."""
    from evodiff_amd.pretrained import OA_DM_640M
    torch.manual_seed(0)
    m, _, tok, _ = OA_DM_640M(random_init=True)
    m.precision = "bf16"
    B, L, n = 64, 256, 4
    order = torch.from_numpy(np.random.default_rng(0).permutation(L)[:n].astype(np.int32))
    uni = torch.from_numpy(np.random.default_rng(1).random((n, B), dtype=np.float32)).to(dev)
    # with one GEMM kernel for every problem size the arithmetic per row is the same at any batch size
    m.set_option("min_tuned_tiles", 0)
    m.set_option("prune", 0)
    m.set_option("incremental", 0)
    big = torch.full((B, L), tok.mask_id, dtype=torch.long, device=dev)
    pb = m.oadm_sample(big, order, uni, R.OADM_DRAW, return_probs=True)
    small = torch.full((2, L), tok.mask_id, dtype=torch.long, device=dev)
    ps = m.oadm_sample(small, order, uni[:, 62:64].contiguous(), R.OADM_DRAW, return_probs=True)
    assert torch.equal(big[62:64], small)
    assert torch.equal(pb[:, 62:64], ps)
    assert int((big != tok.mask_id).sum()) == B * n
    # default kernel choice (small problems on the small-tile kernel): same softmax rows within bf16 noise
    m.set_option("min_tuned_tiles", 130)
    m.set_option("prune", 1)
    big2 = torch.full((B, L), tok.mask_id, dtype=torch.long, device=dev)
    pb2 = m.oadm_sample(big2, order[:1], uni[:1], R.OADM_DRAW, return_probs=True)
    assert (pb2[0] - pb[0]).abs().max() < 2e-2


def test_d3pm_640m_steps_vs_oracle_and_full_batch(dev):
    """BASELINE config 3 (D3PM_BLOSUM_640M, B=64, L=256): single steps at t = 499, 250 and 1 against the oracle at
    B=2 (fp32 posterior 2e-5; bf16 draws outside the margin), then the full batch: rows of a 64-batch bf16 run
    equal the same rows sampled in a batch of 2 (sequences are independent)."""
    from evodiff_amd.utils import DEFAULT_BLOSUM, Tokenizer
    cfg = R.MODEL_CONFIGS["640M"]
    oracle = R.build_model(cfg, 500, 3)
    m = product_model(oracle, cfg, timesteps=500, precision="fp32")
    tok = Tokenizer(path_to_blosum=DEFAULT_BLOSUM, sequences=True)
    Qbar, Q = tok.q_blosum_schedule(timesteps=500)
    B, L, K = 64, 256, tok.K
    rng = np.random.default_rng(2)
    start = torch.from_numpy(rng.integers(0, K, (B, L)))
    uni = torch.from_numpy(np.random.default_rng(3).random((500, B, L), dtype=np.float32))
    small, uni_small = start[62:64].contiguous(), uni[:, 62:64].contiguous()
    for t in (499, 250, 1):
        ref_tok, _, ref_post, margin = R.d3pm_step(oracle, small, t, Q, Qbar, uni_small[499 - t], K,
                                                   u_final=uni_small[499])
        for precision, ptol, mtol in (("fp32", 2e-5, DRAW_MARGIN_FP32), ("bf16", 5e-2, DRAW_MARGIN_BF16)):
            m.precision = precision
            sample = small.clone().to(dev)
            post = m.d3pm_sample(sample, Q, Qbar, uni_small, K, t_start=t, n_steps=1, return_post=True)
            assert np.abs(post[0].cpu().numpy() - ref_post.numpy()).max() < ptol
            n_safe = _check_draws(sample.cpu().numpy(), ref_tok.numpy(), margin.numpy(), mtol)
            assert n_safe > 0.5 * small.numel()
    m.precision = "bf16"
    m.set_option("min_tuned_tiles", 0)       # one GEMM kernel at both batch sizes: the same arithmetic per row
    big = start.clone().to(dev)
    m.d3pm_sample(big, Q, Qbar, uni, K, t_start=3, n_steps=3)
    two = small.clone().to(dev)
    m.d3pm_sample(two, Q, Qbar, uni_small, K, t_start=3, n_steps=3)
    assert torch.equal(big[62:64], two)
    assert int(big.max()) < K - 6            # the chain ended at t == 1: canonical amino acids only


def test_640m_length_512_two_tiles_per_sequence(dev):
    """Geometry of BASELINE config 4 (L=512: two 256-row tiles per sequence, dilation halos crossing the tile
    boundary, pruning down to one row of 512): OA-DM steps against the oracle at B=2."""
    cfg = R.MODEL_CONFIGS["640M"]
    oracle = R.build_model(cfg, None, 4)
    m = product_model(oracle, cfg, precision="fp32")
    B, L = 2, 512
    rng = np.random.default_rng(5)
    state = torch.full((B, L), R.MASK_ID, dtype=torch.long)
    state[:, rng.permutation(L)[: L // 3]] = torch.from_numpy(rng.integers(0, 20, (B, L // 3)))
    zeros = torch.zeros(B, dtype=torch.long, device=dev)
    for pos, u in ((0, 0.3), (255, 0.9), (256, 0.5), (511, 0.1)):
        uu = torch.full((B,), u)
        nxt, logits, p, margin = R.oadm_step(oracle, state, pos, uu)
        for precision, tol, mtol in (("fp32", FP32_TOL, DRAW_MARGIN_FP32), ("bf16", BF16_TOL, DRAW_MARGIN_BF16)):
            m.precision = precision
            if pos == 256:
                assert rel_err(m(state.to(dev), zeros).cpu().numpy(), logits.numpy()) < tol
            sample = state.clone().to(dev)
            probs = m.oadm_sample(sample, torch.tensor([pos]), uu.view(1, B).to(dev), R.OADM_DRAW, return_probs=True)
            assert np.abs(probs[0].cpu().numpy() - p.numpy()).max() < (1e-5 if precision == "fp32" else 5e-2)
            safe = margin.numpy() > mtol
            assert np.array_equal(sample[:, pos].cpu().numpy()[safe], nxt[:, pos].numpy()[safe])
        state = nxt


def test_38m_config5_full_size_batch_invariance(dev):
    """BASELINE config 5 (OA_DM_38M, B=1024, L=1024, bf16; N-tile 256 kernels, four tiles per sequence): rows of the
    full batch equal the same rows sampled in a batch of 4, and that small batch follows the oracle."""
    from evodiff_amd.pretrained import OA_DM_38M
    torch.manual_seed(0)
    m, _, tok, _ = OA_DM_38M(random_init=True)
    m.precision = "bf16"
    m.set_option("min_tuned_tiles", 0)
    m.set_option("prune", 0)
    m.set_option("incremental", 0)            # (the number of incremental leading layers depends on the batch size)
    B, L, n = 1024, 1024, 2
    order = torch.from_numpy(np.random.default_rng(0).permutation(L)[:n].astype(np.int32))
    uni = torch.from_numpy(np.random.default_rng(1).random((n, B), dtype=np.float32)).to(dev)
    big = torch.full((B, L), tok.mask_id, dtype=torch.long, device=dev)
    pb = m.oadm_sample(big, order, uni, R.OADM_DRAW, return_probs=True)
    small = torch.full((4, L), tok.mask_id, dtype=torch.long, device=dev)
    ps = m.oadm_sample(small, order, uni[:, 1020:1024].contiguous(), R.OADM_DRAW, return_probs=True)
    assert torch.equal(big[1020:1024], small)
    assert torch.equal(pb[:, 1020:1024], ps)
    assert int((big != tok.mask_id).sum()) == B * n
    # the small batch against the oracle (same seeded default init), first step, default execution options
    m.set_option("min_tuned_tiles", 130)
    m.set_option("prune", 1)
    m.set_option("incremental", 1)
    torch.manual_seed(0)
    oracle = R.build_model(R.MODEL_CONFIGS["38M"], None, None)
    oracle.load_state_dict(m.state_dict())
    state = torch.full((4, L), R.MASK_ID, dtype=torch.long)
    _, _, p_ref, _ = R.oadm_step(oracle, state, int(order[0]), uni[0, 1020:1024].cpu())
    again = state.clone().to(dev)
    p_got = m.oadm_sample(again, order[:1], uni[:1, 1020:1024].contiguous(), R.OADM_DRAW, return_probs=True)
    assert np.abs(p_got[0].cpu().numpy() - p_ref.numpy()).max() < 5e-2


def test_conditional_callers_against_oracle(dev):
    """inpaint_simple / generate_scaffold are the OA-DM loop with a partially filled start and an
    order over the masked positions (B = 1: the weight-streaming regime; repeated positions when batch_size > 1).
    fp32 tokens must equal the oracle's loop on the same start, order and uniforms."""
    from evodiff_amd.conditional_generation import generate_scaffold, get_intervals, inpaint_simple
    from evodiff_amd.utils import Tokenizer
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "oadm_trace.npz"))
    oracle = R.build_model(TINY_WIDE, None, 12)
    with torch.no_grad():
        oracle.decoder.conv.weight.mul_(float(g["decoder_scale"]))
    m = product_model(oracle, TINY_WIDE, precision="fp32")
    tok = Tokenizer()
    rng = np.random.default_rng(7)
    # --- inpainting a region of a 120-residue sequence
    seq = "".join(rng.choice(list("ACDEFGHIKLMNPQRSTVWY"), 120))
    s0, s1 = 37, 71
    order = rng.permutation(np.arange(s0, s1))
    uni = torch.from_numpy(rng.random((len(order), 1), dtype=np.float32))
    start = torch.tensor(tok.tokenizeMSA(seq[:s0] + "#" * (s1 - s0) + seq[s1:]), dtype=torch.long).unsqueeze(0)
    trace = []
    ref = R.generate_oaardm_ref(oracle, 120, 1, order, uni, trace=trace, start=start)
    sample, full, region = inpaint_simple(m, seq, s0, s1, tokenizer=tok, device=dev, order=order, uniforms=uni)
    margins = np.array([float(t["margin"].min()) for t in trace])
    assert margins.min() > DRAW_MARGIN_FP32, "seed gives a boundary-ambiguous draw; pick another"
    assert np.array_equal(sample.cpu().numpy(), ref[0].numpy())
    assert full[:s0] == seq[:s0] and full[s1:] == seq[s1:] and region == full[s0:s1] and "#" not in full
    # --- scaffolding a motif, two sequences: every masked position is visited twice (reference quirk)
    motif, scaffold_length, new_start = "HELLKAGH", 40, 11
    L = scaffold_length + len(motif)
    masked = np.array([p for p in range(L) if not (new_start <= p < new_start + len(motif))])
    order = rng.permutation(np.concatenate([masked, masked]))
    uni = torch.from_numpy(rng.random((len(order), 2), dtype=np.float32))
    start = torch.full((2, L), R.MASK_ID, dtype=torch.long)
    start[:, new_start:new_start + len(motif)] = torch.tensor(tok.tokenize((motif,)))
    trace = []
    ref = R.generate_oaardm_ref(oracle, L, 2, order, uni, trace=trace, start=start)
    strings, starts, ends = generate_scaffold(m, "1abc", [0], [len(motif) - 1], scaffold_length, "/nowhere/", tok,
                                              batch_size=2, device=dev, motif_seq=motif, new_start=new_start,
                                              order=order, uniforms=uni)
    assert (starts, ends) == ([new_start], [new_start + len(motif) - 1])
    assert all(s[new_start:new_start + len(motif)] == motif and "#" not in s for s in strings)
    assert min(float(t["margin"].min()) for t in trace) > DRAW_MARGIN_FP32, "seed gives a boundary-ambiguous draw"
    assert strings == [tok.untokenize(r) for r in ref]
    assert get_intervals([3, 4, 5, 9, 10, 20, 21]) == ([3, 9, 20], [5, 10, 21])


def _sharp_oracle(cfg, seed, scale):
    oracle = R.build_model(cfg, None, seed)
    with torch.no_grad():
        oracle.decoder.conv.weight.mul_(float(scale))
    return oracle


def test_repetition_penalty_against_reference_trace(golden_dir, dev):
    """generate_oaardm(penalty=...) (generate.py:202-210): the fixture is the reference's own loop run verbatim with
    pre-drawn uniforms (oracle/make_golden_r3.py; 50 re-draws in 160 draws); every draw's margin there is beyond
    DRAW_MARGIN_FP32, so the fp32 path must reproduce the final tokens and strings exactly."""
    from evodiff_amd.generate import generate_oaardm
    from evodiff_amd.utils import Tokenizer
    g = np.load(os.path.join(golden_dir, "oadm_penalty_trace.npz"))
    assert min(float(g["first_margin"].min()), float(g["redraw_margin"].min())) > DRAW_MARGIN_FP32
    m = product_model(_sharp_oracle(TINY_WIDE, 12, g["decoder_scale"]), TINY_WIDE, precision="fp32")
    B, L = g["final"].shape
    sample, strings = generate_oaardm(m, Tokenizer(), L, penalty=float(g["penalty"]), batch_size=B, device=dev,
                                      order=g["order"], uniforms=torch.from_numpy(g["uniforms"]))
    assert np.array_equal(sample.cpu().numpy(), g["final"])
    assert strings == [str(x) for x in g["strings"]]
    # and without the penalty the same uniforms give other sequences (the branch really ran)
    plain, _ = generate_oaardm(m, Tokenizer(), L, batch_size=B, device=dev, order=g["order"],
                               uniforms=torch.from_numpy(g["uniforms"][:L]))
    assert not np.array_equal(plain.cpu().numpy(), g["final"])


def test_inpaint_and_scaffold_against_reference_trace(golden_dir, dev):
    """inpaint (two sequences of different length, conditional_generation.py:628-675) and generate_scaffold with the
    reference's positional signature (:475-476; a two-segment motif, batch 2) against the reference's own loops run
    verbatim (oracle/make_golden_r3.py), fp32, every margin beyond DRAW_MARGIN_FP32."""
    from evodiff_amd.conditional_generation import generate_scaffold, inpaint
    from evodiff_amd.utils import Tokenizer
    g = np.load(os.path.join(golden_dir, "conditional_trace.npz"))
    assert min(float(g["inpaint_margin"].min()), float(g["scaffold_margin"].min())) > DRAW_MARGIN_FP32
    m = product_model(_sharp_oracle(TINY_WIDE, 12, g["decoder_scale"]), TINY_WIDE, precision="fp32")
    tok = Tokenizer()
    seqs = [str(x) for x in g["inpaint_seqs"]]
    starts, ends = g["inpaint_starts"].tolist(), g["inpaint_ends"].tolist()
    tokenized = [torch.tensor(tok.tokenizeMSA(x)) for x in seqs]
    res = inpaint(m, tokenized, starts, ends, seqs, tokenizer=tok, device=dev,
                  orders=[g["inpaint_order0"], g["inpaint_order1"]],
                  uniforms=[torch.from_numpy(g["inpaint_uni0"]), torch.from_numpy(g["inpaint_uni1"])])
    assert [x[0] for x in res[0]] == [str(x) for x in g["inpaint_out_seqs"]]
    assert res[1] == [[x] for x in seqs]
    assert res[2] == [str(x) for x in g["inpaint_out_idrs"]] and res[3] == [str(x) for x in g["inpaint_ref_idrs"]]
    assert (res[4], res[5]) == (starts, ends)
    # generate_scaffold, called exactly as the reference is (PDB id, motif segment indices, data directory): the motif
    # string stands in for the PDB file
    strings, s_idx, e_idx = generate_scaffold(m, "XXXX", [0, 7], [4, 9], int(g["scaffold_length"]), "/nowhere/", tok,
                                              batch_size=2, device=dev, motif_seq=str(g["scaffold_motif"]),
                                              new_start=int(g["scaffold_new_start"]), order=g["scaffold_order"],
                                              uniforms=torch.from_numpy(g["scaffold_uni"]))
    assert strings == [str(x) for x in g["scaffold_strings"]]
    assert (s_idx, e_idx) == (g["scaffold_start_idxs"].tolist(), g["scaffold_end_idxs"].tolist())
    with pytest.raises(FileNotFoundError, match="does not download"):
        generate_scaffold(m, "XXXX", [0, 7], [4, 9], 30, "/nowhere/", tok, batch_size=1, device=dev)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_autoreg_scaffold_against_reference_trace(golden_dir, dev, precision):
    """generate_autoreg_scaffold (conditional_generation.py:515-576) on ed_ar_sample_scaffold: motif segments spliced
    into the growing sequence, draws over [:stop_id] up to the last motif residue and over the whole alphabet after it,
    cut at STOP.  fp32: the strings of the reference's own loop (run verbatim on the oracle's causal block; margins beyond
    DRAW_MARGIN_FP32).  bf16: per draw against the oracle wherever its margin allows, up to the first draw that does
    not."""
    from evodiff_amd.conditional_generation import generate_autoreg_scaffold
    from evodiff_amd.utils import Tokenizer
    from evodiff_amd.constants import ALL_AAS, PAD, PROTEIN_ALPHABET
    g = np.load(os.path.join(golden_dir, "conditional_trace.npz"))
    alphabet = str(g["ar_alphabet"])
    assert alphabet == PROTEIN_ALPHABET
    cfg = dict(TINY_WIDE, n_tokens=30, causal=True)
    oracle = R.build_model(cfg, None, int(g["ar_seed"]))
    with torch.no_grad():
        oracle.decoder.conv.weight.mul_(float(g["decoder_scale"]))
        oracle.decoder.conv.bias[alphabet.index("*")] += float(g["ar_stop_bias"])
    m = product_model(oracle, cfg, precision=precision)
    tok = Tokenizer(protein_alphabet=PROTEIN_ALPHABET, all_aas=ALL_AAS, pad=PAD)
    motif, n_sc = str(g["ar_motif"]), int(g["ar_scaffold_length"])
    exact = 0
    for r in range(len(g["ar_new_start"])):
        uni = torch.from_numpy(g["ar_uni"][r]).reshape(-1, 1)
        strings, s_idx, e_idx = generate_autoreg_scaffold(m, "XXXX", [0, 4], [2, 6], n_sc, "/nowhere/", tok, batch_size=1,
                                                          device=dev, motif_seq=motif, new_start=int(g["ar_new_start"][r]),
                                                          uniforms=uni)
        assert (s_idx, e_idx) == (g[f"ar{r}_start_idxs"].tolist(), g[f"ar{r}_end_idxs"].tolist())
        want = str(g["ar_strings"][r])
        margin = g[f"ar{r}_margin"]
        tol = DRAW_MARGIN_FP32 if precision == "fp32" else DRAW_MARGIN_BF16
        if margin.min() > tol:
            assert strings[0] == want, r
            exact += 1
        else:                       # comparable up to the first boundary-ambiguous draw: its output position
            ns = int(g["ar_new_start"][r])
            given = np.zeros(n_sc + len(motif), bool)
            given[ns:ns + len(motif)] = np.array(list(motif)) != "#"
            drawn_pos = np.flatnonzero(~given)[: len(margin)]
            first_bad = int(drawn_pos[np.argmax(margin <= tol)])
            assert strings[0][:first_bad] == want[:first_bad], r
    if precision == "fp32":
        assert exact == len(g["ar_new_start"])
    # a batch of scaffolds around one placement: row b equals the single-sequence call with the same uniforms
    uni = torch.from_numpy(np.random.default_rng(3).random((n_sc + len(motif), 3), dtype=np.float32))
    many, _, _ = generate_autoreg_scaffold(m, "XXXX", [0, 4], [2, 6], n_sc, "/nowhere/", tok, batch_size=3, device=dev,
                                           motif_seq=motif, new_start=5, uniforms=uni)
    one, _, _ = generate_autoreg_scaffold(m, "XXXX", [0, 4], [2, 6], n_sc, "/nowhere/", tok, batch_size=1, device=dev,
                                          motif_seq=motif, new_start=5, uniforms=uni[:, 1:2].contiguous())
    assert all(s[5:8] == "WKL" and s[9:12] == "PEN" for s in many if len(s) >= 12)
    if precision == "fp32":         # (in bf16 the K-loop cut of the small problems depends on the batch size)
        assert many[1] == one[0]


def test_errors_are_raised_not_faults(dev):
    from evodiff_amd import _lib
    oracle = R.build_model(TINY, None, 1)
    m = product_model(oracle, TINY)
    with pytest.raises(_lib.EvoDiffHipError):
        m(torch.zeros((1, 4), dtype=torch.long), torch.zeros(1, dtype=torch.long))          # CPU tensors
    with pytest.raises(IndexError):
        m(torch.full((1, 4), 31, dtype=torch.long, device=dev), torch.zeros(1, dtype=torch.long, device=dev))
    sample = torch.full((2, 8), R.MASK_ID, dtype=torch.long, device=dev)
    with pytest.raises(_lib.EvoDiffHipError):
        m.oadm_sample(sample, torch.tensor([8]), torch.zeros((1, 2), device=dev), R.OADM_DRAW)   # position out of range
    with pytest.raises((ValueError, _lib.EvoDiffHipError)):
        m.d3pm_sample(sample, torch.zeros(1), torch.zeros(1), torch.zeros(1), 26)               # no timesteps


def test_device_side_validation_reports_on_the_next_call(dev):
    """The C ABI validates device data on the device: no copy to the host and no synchronisation
    per call.  Out-of-range positions / tokens / targets are clamped (no fault, nothing indexed outside a table) and
    the NEXT call - or ed_model_check_errors - reports them and clears the report."""
    from evodiff_amd import _lib
    oracle = R.build_model(TINY, None, 1)
    m = product_model(oracle, TINY)
    B, L = 2, 8
    uni = torch.full((1, B), 0.5, device=dev)
    sample = torch.full((B, L), R.MASK_ID, dtype=torch.long, device=dev)
    m.oadm_sample(sample, torch.tensor([L], device=dev), uni, R.OADM_DRAW)         # order on the GPU: not checked by Python
    with pytest.raises(_lib.EvoDiffHipError, match="order positions"):
        m.check_errors()
    assert int((sample[:, :L - 1] != R.MASK_ID).sum()) == 0 and int((sample[:, L - 1] < R.OADM_DRAW).sum()) == B  # clamped to L-1
    m.check_errors()                                                                # reported once, then clear
    good = torch.zeros((B, L), dtype=torch.long, device=dev)
    bad = good.clone()
    bad[1, 3] = 31
    t = torch.zeros(B, dtype=torch.long, device=dev)
    want = m(torch.where(bad == 31, torch.full_like(bad, 30), bad), t)
    got = m(bad, t, validate=False)                                                 # embedded as id n_tokens - 1
    assert torch.equal(got, want)
    with pytest.raises(_lib.EvoDiffHipError, match="token ids"):
        m(good, t)                                                                  # the next call reports it ...
    assert torch.isfinite(m(good, t)).all()                                         # ... once
    m.score_nll(good, t, torch.full_like(good, 25), n_classes=21, validate=False)
    with pytest.raises(_lib.EvoDiffHipError, match="targets"):
        m.check_errors()
    # the sampler loops feed the model its own samples without the up-front check: no error pending afterwards
    from evodiff_amd.generate import generate_oaardm_order_opt
    from evodiff_amd.utils import Tokenizer
    generate_oaardm_order_opt(m, Tokenizer(), 12, batch_size=2, device=dev)
    m.check_errors()


def test_torch_operators_equal_the_ctypes_binding(dev):
    """torch.ops.evodiff_amd.* (evodiff_amd/csrc/torch_ops.cpp, the thin PyTorch-ROCm C++ extension of north_star) call
    the same C ABI as the ctypes binding: results are bit-identical, errors surface as RuntimeError."""
    from evodiff_amd import torch_ops
    from evodiff_amd.utils import Tokenizer
    ops = torch_ops.load()
    oracle = R.build_model(TINY_WIDE, None, 12)
    m = product_model(oracle, TINY_WIDE, precision="bf16")
    h = torch_ops.handle_of(m, dev)
    B, L = 3, 70
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.integers(0, 31, (B, L))).to(dev)
    t = torch.zeros(B, dtype=torch.long, device=dev)
    assert torch.equal(ops.forward(h, x, t, 31, 1), m(x, t))
    order = torch.from_numpy(rng.permutation(L).astype(np.int32)).to(dev)
    uni = torch.from_numpy(rng.random((L, B), dtype=np.float32)).to(dev)
    a = torch.full((B, L), R.MASK_ID, dtype=torch.long, device=dev)
    b = a.clone()
    m.oadm_sample(a, order, uni, R.OADM_DRAW)
    out = ops.oadm_sample_(h, b, order, uni, R.OADM_DRAW, 1, True)
    assert out.data_ptr() == b.data_ptr() and torch.equal(a, b) and int((a == R.MASK_ID).sum()) == 0
    tgt = torch.from_numpy(rng.integers(0, 21, (B, L))).to(dev)
    assert torch.equal(ops.score_nll(h, x, t, tgt, None, 31, 1), m.score_nll(x, t, tgt))
    with pytest.raises(RuntimeError, match="n_draw"):
        ops.oadm_sample_(h, b, order, uni, 99, 1, True)
    # D3PM through the operator
    oracle = R.build_model(TINY, 500, 13)
    m = product_model(oracle, TINY, timesteps=500, precision="fp32")
    h = torch_ops.handle_of(m, dev)
    tok = Tokenizer(sequences=True)
    Qbar, Q = tok.q_random_schedule(timesteps=500)
    Q, Qbar = Q.to(dev).double().contiguous(), Qbar.to(dev).double().contiguous()
    uni = torch.from_numpy(rng.random((500, 2, 20), dtype=np.float32)).to(dev)
    a = torch.from_numpy(rng.integers(0, tok.K, (2, 20))).to(dev)
    b = a.clone()
    m.d3pm_sample(a, Q, Qbar, uni, tok.K, t_start=499, n_steps=5)
    ops.d3pm_sample_(h, b, Q, Qbar, uni, tok.K, 499, 5, 0, True)
    assert torch.equal(a, b)


def test_order_opt_sampler_against_reference_trace(golden_dir, dev):
    """generate_oaardm_order_opt (generate.py:151-177), fp32, against the trace the reference's own
    source produced: the same positions are picked and the same tokens drawn wherever neither the argmax nor the
    draw is a near tie."""
    from evodiff_amd.generate import generate_oaardm_order_opt
    from evodiff_amd.utils import Tokenizer
    g = np.load(os.path.join(golden_dir, "order_opt_trace.npz"))
    oracle = R.build_model(TINY_WIDE, None, 12)
    with torch.no_grad():
        oracle.decoder.conv.weight.mul_(float(g["decoder_scale"]))
    L = len(g["uniforms"])
    trace = []
    R.generate_oaardm_order_opt_ref(oracle, L, 2, torch.from_numpy(g["uniforms"]), trace=trace)
    assert min(t["gap"] for t in trace) > 1e-4 and min(t["margin"] for t in trace) > DRAW_MARGIN_FP32, \
        "fixture has a near tie: regenerate it with another seed"
    m = product_model(oracle, TINY_WIDE, precision="fp32")
    sample, strings = generate_oaardm_order_opt(m, Tokenizer(), L, batch_size=2, device=dev,
                                                uniforms=torch.from_numpy(g["uniforms"]))
    assert np.array_equal(sample.cpu().numpy(), g["final"])
    assert strings == [str(s) for s in g["strings"]]


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-4), ("bf16", BF16_TOL)])
def test_scoring_against_reference_losses(golden_dir, dev, precision, tol):
    """sum_nll_mask (analysis/sequence_perp.py:101-156) through ed_score_nll against the losses the
    reference's own collaters + losses produced (fixture scoring.npz); tolerance relative to the summed NLL."""
    from evodiff_amd.sequence_perp import sum_nll_mask
    from evodiff_amd.utils import Tokenizer
    g = np.load(os.path.join(golden_dir, "scoring.npz"))
    oracle = R.build_model(TINY_WIDE, None, 12)
    with torch.no_grad():
        oracle.decoder.conv.weight.mul_(float(g["decoder_scale"]))
    m = product_model(oracle, TINY_WIDE, precision=precision)
    for i in range(3):
        batch = tuple(torch.from_numpy(g[f"oa{i}_{k}"]) for k in ("src", "timestep", "tgt", "mask"))
        frac, nll, tokens = sum_nll_mask(None, (m, None, Tokenizer(), "mask"), device=dev, collated=batch)
        ref = float(g[f"oa{i}_nll"])
        assert tokens == int(g[f"oa{i}_tokens"]) and abs(nll - ref) <= tol * abs(ref), (i, nll, ref)
    oracle = R.build_model(TINY, 500, 13)
    with torch.no_grad():
        oracle.decoder.conv.weight.mul_(float(g["decoder_scale"]))
    m = product_model(oracle, TINY, timesteps=500, precision=precision)
    for i in range(2):
        batch = (torch.from_numpy(g[f"d3{i}_src"]), None, torch.from_numpy(g[f"d3{i}_timestep"]),
                 torch.from_numpy(g[f"d3{i}_tgt"]))
        t_out, nll, tokens = sum_nll_mask(None, (m, None, Tokenizer(sequences=True), "d3pm"), device=dev, collated=batch)
        ref = float(g[f"d3{i}_nll"])
        assert tokens == batch[0].shape[1] and int(t_out) == int(batch[2]) and abs(nll - ref) <= tol * abs(ref)
    # per-position values against the oracle's log-softmax, and the argument checks of the entry point
    src, tgt = torch.from_numpy(g["oab_src"]).to(dev), torch.from_numpy(g["oab_tgt"]).to(dev)
    with torch.no_grad():
        oracle = R.build_model(TINY_WIDE, None, 12)
        ref = torch.nn.functional.cross_entropy(oracle(src.cpu(), torch.zeros(3, dtype=torch.long)).transpose(1, 2),
                                                tgt.cpu(), reduction="none")
    m = product_model(oracle, TINY_WIDE, precision="fp32")
    got = m.score_nll(src, torch.zeros(3, dtype=torch.long), tgt)
    assert rel_err(got.cpu().numpy(), ref.numpy()) < 1e-4
    sel = torch.from_numpy(g["oab_mask"]).to(dev)
    got = m.score_nll(src, torch.zeros(3, dtype=torch.long), tgt, select=sel)
    assert rel_err(got.cpu().numpy(), (ref * sel.cpu()).numpy()) < 1e-4 and float(got[sel == 0].abs().max()) == 0.0
    from evodiff_amd._lib import EvoDiffHipError
    with pytest.raises(EvoDiffHipError, match="targets"):
        m.score_nll(src, torch.zeros(3, dtype=torch.long), tgt, n_classes=21)   # pad / mask ids are >= 21


@pytest.mark.parametrize("base,B,L", [(MID, 2, 37), (MID, 3, 300), (TINY_WIDE, 1, 513), (BIG512, 3, 256),
                                      (BIG640, 4, 512), (BIG640, 16, 130), (BIG640, 64, 256)])
def test_causal_forward_vs_oracle(dev, base, B, L):
    """the LR-AR / CARP variants (pretrained.py:164-205): 30-token alphabet, and causal=True =>
    MaskedCausalConv1d (all taps look back).  Every convolution kernel family takes the causal form: the small-tile
    kernel, the per-tap tuned kernel (dilation >= 32) and the windowed one (dilation <= 16), 128- and 256-row tiles.
    [recalled-upstream: parity of the causal convolution itself is unpinned, the oracle restates it.]"""
    cfg = dict(base, n_tokens=30, causal=True)
    oracle = R.build_model(cfg, None, 31)
    m = product_model(oracle, cfg)
    x = torch.from_numpy(np.random.default_rng(B * 1000 + L).integers(0, 30, (B, L)))
    with torch.no_grad():
        ref = oracle(x, torch.zeros(B, dtype=torch.long)).numpy()
    t = torch.zeros(B, dtype=torch.long, device=dev)
    assert ref.shape == (B, L, 30)
    assert rel_err(m(x.to(dev), t).cpu().numpy(), ref) < FP32_TOL
    m.precision = "bf16"
    m.set_option("min_tuned_tiles", 0)            # tuned kernels wherever the shape allows
    got = m(x.to(dev), t)
    assert rel_err(got.cpu().numpy(), ref) < BF16_TOL
    # causality: changing the tokens after position p leaves the logits up to p bit-identical
    p = L // 2
    x2 = x.clone()
    x2[:, p + 1:] = (x2[:, p + 1:] + 7) % 30
    got2 = m(x2.to(dev), t)
    assert torch.equal(got[:, :p + 1], got2[:, :p + 1]) and not torch.equal(got[:, p + 1:], got2[:, p + 1:])


@pytest.mark.parametrize("base,precision,max_len", [(TINY_WIDE, "fp32", 48), (TINY, "fp32", 300), (TINY_WIDE, "bf16", 48),
                                                    (BIG640, "bf16", 24)])   # wide enough for every GEMM's K loop to be cut
def test_autoregressive_sampler_vs_oracle(dev, base, precision, max_len):
    """generate_autoreg (generate.py:217-256) on a causal model.  The oracle re-runs the whole
    growing prefix at every step like the reference; ed_ar_sample computes one new row per layer from cached
    convolution inputs (max_len = 300 reaches back over every dilation of the schedule).  Three sequences side by
    side, each checked against its own oracle run: per-step probabilities, and tokens up to the first draw that is
    closer to a category boundary than the margin."""
    from evodiff_amd.generate import generate_autoreg
    from evodiff_amd.utils import Tokenizer
    from evodiff_amd.constants import ALL_AAS, PAD, PROTEIN_ALPHABET
    cfg = dict(base, n_tokens=30, causal=True)
    oracle = R.build_model(cfg, None, 41)
    with torch.no_grad():
        oracle.decoder.conv.weight.mul_(4.0)
    m = product_model(oracle, cfg, precision=precision)
    tok = Tokenizer(protein_alphabet=PROTEIN_ALPHABET, all_aas=ALL_AAS, pad=PAD)
    assert (tok.start_id, tok.stop_id, tok.mask_id) == (29, 26, 28)
    n = 3
    uni = torch.from_numpy(np.random.default_rng(17).random((n, max_len), dtype=np.float32))
    sample = torch.full((n, max_len + 1), tok.start_id, dtype=torch.long, device=dev)
    probs = m.ar_sample(sample, uni.t().contiguous().to(dev), n_steps=max_len, return_probs=True).cpu()
    margin = DRAW_MARGIN_FP32 if precision == "fp32" else DRAW_MARGIN_BF16
    p_tol = 1e-5 if precision == "fp32" else 5e-2
    seqs, _ = generate_autoreg(m, tok, samples=n, batch_size=2, max_seq_len=max_len, device=dev, uniforms=uni)
    checked = 0
    for b in range(n):
        trace = []
        # the per-step check runs past STOP (the GPU loop does too); the reference's own stopping rule gives `ref`
        R.generate_autoreg_ref(oracle, tok.start_id, -1, max_len, uni[b], trace=trace)
        ref = R.generate_autoreg_ref(oracle, tok.start_id, tok.stop_id, max_len, uni[b])
        safe = len(trace)
        for s, tr in enumerate(trace):
            assert float((probs[s, b] - tr["p"]).abs().max()) < p_tol, (b, s)
            if tr["margin"] <= margin:
                safe = s          # this draw may legitimately differ, and everything after it follows another path
                break
            assert int(sample[b, s + 1]) == tr["tok"], (b, s)
        checked += safe
        if safe == len(trace):    # whole sequence comparable: the cut at STOP and the dropped last token too
            assert torch.equal(seqs[b].cpu(), ref), b
    if precision == "fp32":
        assert checked >= max_len
    # teacher-forced along the GPU's own path (one causal oracle forward over the finished rows gives every step's
    # distribution): per-step probabilities, and each token is the inverse-CDF draw from the GPU's probabilities
    with torch.no_grad():
        p_ref = torch.softmax(oracle(sample.cpu(), torch.zeros(n, dtype=torch.long)), dim=-1)
    for b in range(n):
        assert float((probs[:, b] - p_ref[b, :max_len]).abs().max()) < p_tol
        idx, mg = R.inverse_cdf(probs[:, b], uni[b])
        ok = mg > 1e-6
        assert torch.equal(idx[ok], sample[b, 1:].cpu()[ok]) and int(ok.sum()) > max_len // 2
    # graph replay and eager launches agree bit for bit; a second call reuses the caches' zero padding
    m.use_graph = False
    again = torch.full((n, max_len + 1), tok.start_id, dtype=torch.long, device=dev)
    m.ar_sample(again, uni.t().contiguous().to(dev), n_steps=max_len)
    assert torch.equal(again, sample)
    with pytest.raises(Exception, match="causal"):
        product_model(R.build_model(dict(base, n_tokens=30), None, 41), dict(base, n_tokens=30)).ar_sample(
            again, uni.t().contiguous().to(dev), n_steps=max_len)


def test_flagship_geometry_free_running_bf16_trajectory_against_oracle(dev):
    """The 640M block geometry at BASELINE config 2's batch (d_model = d_hidden = 1280, 64 sequences of 256: one tile per
    CU, the 20-step PFF2 with the residual inside its K loop, the windowed convolution at every dilation, LayerNorms in
    the epilogues, incremental leading layers with the cached LayerNorm 1, pruned trailing layers) sampling FREE-RUNNING
    in bf16 under graph replay - and every draw of three of the sequences checked against the oracle, teacher-forced
    along the GPU's own trajectory: the state before step s is the final sample with positions order[s:] masked again,
    the oracle's fp32 softmax row at order[s] must agree with the GPU's draw wherever the uniform is not within
    DRAW_MARGIN_BF16 of a category boundary."""
    cfg, B, L, n = WIDE1280, 64, 256, 12
    oracle = R.build_model(cfg, None, 91)
    with torch.no_grad():
        oracle.decoder.conv.weight.mul_(8.0)       # peaky softmax rows: the draws are not all near-uniform
    m = product_model(oracle, cfg, precision="bf16")
    m.use_graph = True
    order = np.random.default_rng(7).permutation(L)[:n].astype(np.int32)
    uni = torch.from_numpy(np.random.default_rng(8).random((n, B), dtype=np.float32))
    sample = torch.full((B, L), R.MASK_ID, dtype=torch.long, device=dev)
    m.oadm_sample(sample, torch.from_numpy(order), uni.to(dev), R.OADM_DRAW)
    m.check_errors()
    final = sample.cpu()
    assert (final[:, order] < R.OADM_DRAW).all() and int((final == R.MASK_ID).sum()) == B * (L - n)
    rows = [0, 31, 63]
    n_safe = 0
    for s in range(n):
        state = final[rows].clone()
        state[:, order[s:]] = R.MASK_ID
        with torch.no_grad():
            logits = oracle(state, torch.zeros(len(rows), dtype=torch.long))[:, int(order[s]), :R.OADM_DRAW]
        idx, margin = R.inverse_cdf(torch.softmax(logits, dim=-1), uni[s, rows])
        n_safe += _check_draws(final[rows, int(order[s])].numpy(), idx.numpy(), margin.numpy(), DRAW_MARGIN_BF16)
    assert n_safe > 0.5 * n * len(rows), "test is vacuous: most draws sit on a category boundary"


@pytest.mark.parametrize("base,B", [(TINY_WIDE, 1), (TINY_WIDE, 5), (BIG640, 8), (MID, 3), (WIDE1280, 2)])
def test_autoregressive_weight_streaming_kernels_against_oracle_and_tile_kernels(dev, base, B):
    """ED_OPT_AR_GEMV (ar_gemv.hip; the option is the largest batch that takes them, 8 here so that every case does):
    autoregressive steps of at most 8 sequences as three weight-streaming launches per block, each applying the LayerNorm in front of its GEMM itself.  Per-step probabilities against the oracle
    (teacher-forced along the GPU's own sequences: one causal forward gives every step's distribution) and against the
    tile-kernel path (same arithmetic up to the order of the fp32 sums and one bf16 rounding less); tokens are the
    inverse-CDF draws from the GPU's own probabilities; graph replay equals eager launches bit for bit; 9 sequences take
    the tile kernels whatever the option says.  Shapes: 256 / 640 / 1280 wide, slim (d_hidden = d_model / 2 = 192)."""
    cfg = dict(base, n_tokens=30, causal=True, n_layers=min(base["n_layers"], 9))
    oracle = R.build_model(cfg, None, 57)
    with torch.no_grad():
        oracle.decoder.conv.weight.mul_(4.0)
    m = product_model(oracle, cfg, precision="bf16")
    n = 40
    uni = torch.from_numpy(np.random.default_rng(B).random((n, B), dtype=np.float32)).to(dev)

    def run(gemv, use_graph=True, rows=B, u=uni):
        m.set_option("ar_gemv", gemv)
        m.use_graph = use_graph
        sample = torch.full((rows, n + 1), 29, dtype=torch.long, device=dev)
        probs = m.ar_sample(sample, u, n_steps=n, return_probs=True)
        return sample.cpu(), probs.cpu()

    tok_g, p_g = run(8)
    tok_t, p_t = run(0)
    m.check_errors()
    with torch.no_grad():
        p_ref = torch.softmax(oracle(tok_g, torch.zeros(B, dtype=torch.long)), dim=-1)[:, :n]      # (B, n, 30)
    assert float((p_g.permute(1, 0, 2) - p_ref).abs().max()) < 5e-2
    for b in range(B):
        idx, mg = R.inverse_cdf(p_g[:, b], uni[:, b].cpu())
        ok = mg > 1e-6
        assert torch.equal(idx[ok], tok_g[b, 1:][ok])
    # the two paths: same distributions within bf16 noise along a common prefix (they part at the first differing draw)
    same = (tok_g == tok_t).all(dim=0).long().cumprod(0)
    k = int(same.sum()) - 1                       # steps whose inputs are equal on both paths
    assert k >= 1 and float((p_g[:k] - p_t[:k]).abs().max()) < 3e-2
    tok_e, p_e = run(8, use_graph=False)
    assert torch.equal(tok_e, tok_g) and torch.equal(p_e, p_g)
    # few workgroups: every workgroup (wave, for one sequence) works through several column groups in passes - the same
    # sums in the same order, so bit-equal
    m.set_option("ar_gemv_wgs", 3)
    tok_w, p_w = run(8)
    m.set_option("ar_gemv_wgs", 0)
    assert torch.equal(tok_w, tok_g) and torch.equal(p_w, p_g)
    # more sequences than the streaming kernels take: the option changes nothing
    u9 = torch.from_numpy(np.random.default_rng(99).random((n, 9), dtype=np.float32)).to(dev)
    a, pa = run(8, rows=9, u=u9)
    b_, pb = run(0, rows=9, u=u9)
    assert torch.equal(a, b_) and torch.equal(pa, pb)


def random_geometry_case(seed, dev):
    """One case of the random-geometry sweep (tools/fuzz_parity.py runs many; six are in the suite below): forward in
    fp32 and bf16 against the oracle; non-causal without timesteps, a free-running fp32 OA-DM loop with default options
    checked draw by draw against the oracle teacher-forced along the GPU's trajectory and bit-equal to the loop with
    pruning / incremental layers / graph replay off; causal, a cached bf16 autoregressive loop checked per step against
    the oracle's causal forward of the sampled sequences.  Returns a one-line summary; asserts on any difference."""
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
    return f"{tag}: fp32 {e32:.1e} bf16 {e16:.1e}{extra}"


@pytest.mark.parametrize("seed", [1, 5, 6, 9, 16, 27])
def test_random_geometries_from_the_parity_sweep(dev, seed):
    """Six cases of the sweep (430 random geometries, all green: profiles/r03_fuzz_parity.txt) kept as regression:
    dilation schedules r = 1 / 4 / 16, ReLU blocks, widths 320 / 384 (no tuned kernel), slim and full hidden layers,
    L = 600 / 255 / 127, causal models with the cached autoregressive loop on the weight-streaming kernels."""
    random_geometry_case(seed, dev)


def test_benchmark_configuration_pinned_to_the_oracle(dev):
    """BASELINE config 2 exactly as bench.py runs it - the real 56-layer OA_DM_640M(random_init=True) under seed 0 with
    bench.py's decoder sharpening, B = 64, L = 256, bf16, DEFAULT execution options (LayerNorms in the epilogues, pruned
    trailing layers, incremental leading layers with the cached LayerNorm 1), graph replay, bench.py's own seeding of the
    order and the uniforms - held to the oracle directly (VERDICT r02 weak point 1):
    (a) ``model(x)`` on a half-filled batch of 64: the logits of three of its rows within BF16_TOL;
    (b) the whole free-running 256-step loop, then three rows x 16 steps spread over the loop (the cache-filling step,
        incremental steps, early / middle / late positions) teacher-forced through the oracle (oracle/spot_check.py): every
        draw outside DRAW_MARGIN_BF16 of a category boundary is the oracle's."""
    import bench
    from evodiff_amd.parallel import batch_order, sequence_uniforms
    from evodiff_amd.pretrained import OA_DM_640M
    from oracle.spot_check import oadm_spot_check, spread_steps
    B, L = 64, 256
    torch.manual_seed(0)
    m, _, tok, _ = OA_DM_640M(random_init=True)
    bench.sharpen_decoder(m, bench.DECODER_SCALE)
    m.precision, m.use_graph = "bf16", True
    oracle = R.build_model(R.MODEL_CONFIGS["640M"], None, None)
    oracle.load_state_dict(m.state_dict())
    order = batch_order(bench.SEED_ORDER, 0, L)
    uni = torch.from_numpy(sequence_uniforms(bench.SEED_UNIFORM, 0, B, (L,)))
    rows = [0, 31, 63]
    # (a) forward
    rng = np.random.default_rng(4)
    x = torch.full((B, L), R.MASK_ID, dtype=torch.long)
    x[:, rng.permutation(L)[: L // 2]] = torch.from_numpy(rng.integers(0, 20, (B, L // 2)))
    got = m(x.to(dev), torch.zeros(B, dtype=torch.long, device=dev))[rows].cpu().numpy()
    with torch.no_grad():
        want = oracle(x[rows], torch.zeros(len(rows), dtype=torch.long)).numpy()
    assert rel_err(got, want) < BF16_TOL
    # (b) the benchmarked loop itself
    sample = torch.full((B, L), tok.mask_id, dtype=torch.long, device=dev)
    m.oadm_sample(sample, torch.from_numpy(order), uni.to(dev), R.OADM_DRAW)
    m.check_errors()
    final = sample.cpu()
    assert int((final == tok.mask_id).sum()) == 0 and int(final.max()) < R.OADM_DRAW
    res = oadm_spot_check(oracle, final, order, uni, rows, spread_steps(L, 16), DRAW_MARGIN_BF16)
    assert res["agree"] == res["draws"], res
    assert res["draws"] >= 0.5 * (res["draws"] + res["near_boundary"]), f"vacuous: {res}"
    # the same call is what bench.py times: equal seeds, equal tokens (its tokens_sha256 covers this batch)
    again = torch.full((B, L), tok.mask_id, dtype=torch.long, device=dev)
    m.oadm_sample(again, torch.from_numpy(order), uni.to(dev), R.OADM_DRAW)
    assert torch.equal(again, sample)


def test_flagship_geometry_d3pm_forward_against_oracle(dev):
    """The same geometry with the D3PM time encoding (per-sequence timesteps through the reshape quirk Q1): bf16 logits of
    the whole batch against the oracle on a few of its sequences - the oracle must see the WHOLE timestep vector, because
    row r of the flattened batch takes y[r % B] (evodiff/model.py:131-140)."""
    cfg, B, L = dict(WIDE1280, n_layers=6), 64, 256
    oracle = R.build_model(cfg, 500, 92)
    m = product_model(oracle, cfg, timesteps=500, precision="bf16")
    rng = np.random.default_rng(9)
    x = torch.from_numpy(rng.integers(0, 26, (B, L)))
    y = torch.from_numpy(rng.integers(0, 500, (B,)))
    got = m(x.to(dev), y.to(dev)).cpu().numpy()
    with torch.no_grad():
        ref = oracle(x, y).numpy()
    assert rel_err(got, ref) < BF16_TOL
    m.check_errors()


def test_in_launch_exchanges_under_uneven_load(dev):
    """The workgroups of a row tile exchange statistics (fused LayerNorm) INSIDE a launch.  An idle chip hides most
    hand-off mistakes, so here other work runs next to them: a second stream keeps launching matrix products of
    changing size, which take CUs away at arbitrary moments and delay some workgroups of a row tile against the others.
    Every forward must still equal the quiet one bit for bit and no bounded wait may run out."""
    cfg, B, L = WIDE1280, 64, 256
    cfg = dict(cfg, n_layers=6)
    oracle = R.build_model(cfg, None, 78)
    m = product_model(oracle, cfg, precision="bf16")
    x = torch.from_numpy(np.random.default_rng(6).integers(0, 31, (B, L))).to(dev)
    t = torch.zeros(B, dtype=torch.long, device=dev)
    quiet = m(x, t).clone()
    side = torch.cuda.Stream()
    mats = [torch.randn((n, n), device=dev, dtype=torch.bfloat16) for n in (512, 1536, 3072, 4096)]
    torch.cuda.synchronize()
    for rep in range(12):
        with torch.cuda.stream(side):
            for k in range(6):
                a = mats[(rep + k) % len(mats)]
                a @ a
        assert torch.equal(m(x, t), quiet), f"rep {rep}"
    torch.cuda.synchronize()
    m.check_errors()


@pytest.mark.parametrize("cfg,B,L", [(BIG640, 4, 512), (BIG640, 16, 130), (BIG640, 64, 256), (BIG512, 9, 300),
                                     (BIG512, 3, 256), (BIG640, 1, 1024), (BIG640, 5, 77),
                                     # 256-row tiles with a ragged last tile per sequence and d_hidden == d_model: the
                                     # largest statistics slab per token (it used to be sized for 128-row tiles)
                                     (BIG640, 40, 300), (TINY_WIDE, 200, 100)])
def test_layernorm_in_gemm_epilogue_matches_separate_launches(dev, cfg, B, L):
    """ED_OPT_FUSE_LN: on the tuned kernels every GEMM of a block applies the LayerNorm + GELU behind it in its own
    epilogue (the column tiles of a row tile exchange per-row statistics inside the launch).  Against separate
    LayerNorm launches only the rounding differs (statistics on the fp32 sums instead of their bf16 roundings): logits
    agree far inside the bf16 tolerance, both stay within it of the oracle, and repeated launches are bit-identical
    (fixed order of every reduction, whichever workgroup arrives first).  Shapes: full and ragged row tiles, 128- and
    256-row tiles, one and two column tiles per row tile, the per-tap convolution form (L > 256, dilation >= 32)."""
    oracle = R.build_model(cfg, None, 21)
    m = product_model(oracle, cfg, precision="bf16")
    m.set_option("min_tuned_tiles", 0)            # tuned kernels wherever the shape allows
    x = torch.from_numpy(np.random.default_rng(B * 1000 + L).integers(0, 31, (B, L)))
    with torch.no_grad():
        ref = oracle(x, torch.zeros(B, dtype=torch.long)).numpy()
    t = torch.zeros(B, dtype=torch.long, device=dev)
    m.set_option("fuse_ln", 0)
    apart = m(x.to(dev), t).cpu().numpy()
    m.set_option("fuse_ln", 1)
    fused = m(x.to(dev), t).cpu().numpy()
    assert rel_err(apart, ref) < BF16_TOL and rel_err(fused, ref) < BF16_TOL
    assert rel_err(fused, apart) < 1e-2
    for _ in range(3):
        assert np.array_equal(m(x.to(dev), t).cpu().numpy(), fused)
    # a free-running pruned + incremental sampling loop under graph replay: many back-to-back launches on the same
    # arrival counters, eager and replayed launches bit-identical
    n = min(L, 16)
    order = torch.from_numpy(np.random.default_rng(L).permutation(L)[:n].astype(np.int32))
    uni = torch.from_numpy(np.random.default_rng(1).random((n, B), dtype=np.float32)).to(dev)
    outs = []
    for use_graph in (True, False, True):
        m.use_graph = use_graph
        sample = torch.full((B, L), R.MASK_ID, dtype=torch.long, device=dev)
        m.oadm_sample(sample, order, uni, R.OADM_DRAW)
        outs.append(sample.cpu().numpy())
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    assert (outs[0][:, order.numpy()] < R.OADM_DRAW).all()
