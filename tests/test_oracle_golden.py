"""The CPU oracle (oracle/bytenet_ref.py) against fixtures produced by the reference's own sources
(oracle/make_golden.py).  No GPU, no /root/reference needed."""
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import bytenet_ref as R

TINY = dict(d_embed=8, d_model=64, n_layers=8, kernel_size=5, r=128, slim=True, activation="gelu")
TINY_WIDE = dict(d_embed=8, d_model=256, n_layers=9, kernel_size=5, r=128, slim=False, activation="gelu")


def sd_digest(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(np.ascontiguousarray(v.detach().numpy()).tobytes())
    return h.hexdigest()


def arr_digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


@pytest.mark.parametrize("tag,cfg,seed", [("tiny", TINY, 11), ("tinywide", TINY_WIDE, 12)])
def test_forward_matches_reference(golden_dir, tag, cfg, seed):
    g = load(golden_dir, f"{tag}_forward.npz")
    m = R.build_model(cfg, None, seed)
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["keys"]]
    assert sd_digest(sd) == str(g["sd_sha"])
    if tag == "tiny":
        for k, v in sd.items():
            assert np.array_equal(v.numpy(), g["sd/" + k]), k
    for L in (7, 100, 300):
        x = torch.from_numpy(g[f"x_L{L}"])
        collect = []
        with torch.no_grad():
            y = m(x, torch.zeros(2, dtype=torch.long), collect=collect)
        assert np.array_equal(y.numpy(), g[f"logits_L{L}"])          # same ops, same machine class
        if g[f"blocks_L{L}"].size:
            assert np.array_equal(torch.stack(collect).numpy(), g[f"blocks_L{L}"])


def test_time_encoding_scramble(golden_dir):
    g = load(golden_dir, "tiny_timeenc.npz")
    m = R.build_model(TINY, 500, 13)
    assert sd_digest(m.state_dict()) == str(g["sd_sha"])
    assert np.array_equal(R.timestep_table(8, 500).numpy(), g["pe"])
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
    with torch.no_grad():
        assert np.array_equal(m.embedder.embed(x, y).numpy(), g["embed"])
        assert np.array_equal(m(x, y).numpy(), g["logits"])
    # the closed form of quirk Q1: row (b, l) gets pe[y[(b*L + l) % B]]
    B, L = x.shape
    idx = (torch.arange(B * L) % B).reshape(B, L)
    pe = R.timestep_table(8, 500)[y][idx]
    with torch.no_grad():
        want = m.embedder.up_embedder(m.embedder.embedder(x) + pe)
    assert torch.allclose(want, torch.from_numpy(g["embed"]), atol=1e-6)


def test_38m_forward_and_param_count(golden_dir):
    g = load(golden_dir, "m38_forward.npz")
    m = R.build_model(R.MODEL_CONFIGS["38M"], None, 0)
    assert sum(p.numel() for p in m.parameters()) == int(g["n_params"]) == 37890327
    assert len(m.state_dict()) == int(g["n_tensors"]) == 199
    assert sd_digest(m.state_dict()) == str(g["sd_sha"])
    with torch.no_grad():
        y = m(torch.from_numpy(g["x"]), torch.zeros(2, dtype=torch.long))
    np.testing.assert_allclose(y.numpy(), g["logits"], rtol=0, atol=2e-6)


def test_schedules(golden_dir):
    g = load(golden_dir, "schedules.npz")
    assert R.MSA_ALPHABET == str(g["alphabet"]) and R.MASK_ID == int(g["mask_id"]) == 28
    assert R.Tokenizer().K == int(g["K_msa"]) == 27
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    blosum = os.path.join(here, "evodiff_amd", "data", "blosum62-special-MSA.mat")
    tb = R.Tokenizer(path_to_blosum=blosum, sequences=True)
    assert tb.K == int(g["K"]) == 26
    np.testing.assert_array_equal(tb.q_blosum().numpy(), g["blosum_q"])
    Qbar, Q = tb.q_blosum_schedule(timesteps=500)
    rows = g["rows"]
    np.testing.assert_array_equal(Q.numpy()[rows], g["blosum_Q"])
    np.testing.assert_array_equal(Qbar.numpy()[rows], g["blosum_Qbar"])
    assert arr_digest(Q.numpy()) == str(g["blosum_Q_sha"])
    assert arr_digest(Qbar.numpy()) == str(g["blosum_Qbar_sha"])
    Ubar, U = R.Tokenizer(sequences=True).q_random_schedule(timesteps=500)
    np.testing.assert_array_equal(U.numpy()[rows], g["uniform_Q"])
    assert arr_digest(U.numpy()) == str(g["uniform_Q_sha"])
    assert arr_digest(Ubar.numpy()) == str(g["uniform_Qbar_sha"])


def test_oadm_trace(golden_dir):
    """generate_oaardm executed by the reference vs the oracle's loop: same order, same uniforms."""
    g = load(golden_dir, "oadm_trace.npz")
    m = R.build_model(TINY_WIDE, None, 12)
    with torch.no_grad():
        m.decoder.conv.weight.mul_(float(g["decoder_scale"]))
    assert sd_digest(m.state_dict()) == str(g["sd_sha"])
    trace = []
    order, uni = g["order"], torch.from_numpy(g["uniforms"])
    out = R.generate_oaardm_ref(m, len(order), 2, order, uni, trace=trace)
    assert np.array_equal(out.numpy(), g["final"])
    assert [R.Tokenizer().untokenize(s) for s in out] == [str(s) for s in g["strings"]]
    np.testing.assert_allclose(torch.stack([t["p"] for t in trace]).numpy(), g["p"], atol=1e-6)
    assert np.array_equal(torch.stack([t["sample"][:, t["pos"]] for t in trace]).numpy(), g["tok"])


@pytest.mark.parametrize("tag", ["blosum", "uniform"])
def test_d3pm_trace(golden_dir, tag):
    g = load(golden_dir, f"d3pm_trace_{tag}.npz")
    m = R.build_model(TINY, 500, 13)
    with torch.no_grad():
        m.decoder.conv.weight.mul_(float(g["decoder_scale"]))
    assert sd_digest(m.state_dict()) == str(g["sd_sha"])
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if tag == "blosum":
        tok = R.Tokenizer(path_to_blosum=os.path.join(here, "evodiff_amd", "data", "blosum62-special-MSA.mat"),
                          sequences=True)
        Qbar, Q = tok.q_blosum_schedule(timesteps=500)
    else:
        tok = R.Tokenizer(sequences=True)
        Qbar, Q = tok.q_random_schedule(timesteps=500)
    trace = []
    out = R.generate_d3pm_ref(m, Q, Qbar, 500, torch.from_numpy(g["start"]), torch.from_numpy(g["uniforms"]),
                              K=tok.K, trace=trace)
    assert np.array_equal(out.numpy(), g["final"])
    toks = np.stack([t["sample"].numpy() for t in trace[:-1]])
    assert np.array_equal(toks, g["toks"])
    for j, s in enumerate(g["keep"]):
        np.testing.assert_allclose(trace[int(s)]["post"].numpy(), g["post"][j], rtol=1e-9, atol=1e-12)
    # t == 1: restricted draw over the first K-6 categories (un-normalised slice of the posterior)
    np.testing.assert_allclose(trace[-1]["post"][..., :tok.K - 6].numpy(), g["last_post"], rtol=1e-9, atol=1e-12)


def test_inverse_cdf_edges():
    p = torch.tensor([[0.25, 0.25, 0.5], [0.0, 1.0, 0.0], [1.0, 0.0, 0.0]])
    idx, _ = R.inverse_cdf(p, torch.tensor([0.0, 0.0, 0.999999]))
    assert idx.tolist() == [0, 1, 0]
    idx, _ = R.inverse_cdf(p, torch.tensor([0.25, 0.5, 0.5]))
    assert idx.tolist() == [1, 1, 0]
    # clamp: u*total == total can only happen through rounding; never index past the end
    idx, _ = R.inverse_cdf(torch.tensor([[0.5, 0.5]]), torch.tensor([1.0]))
    assert idx.tolist() == [1]


def test_f_eff_matches_survey():
    # 640M L=256 → 1147.5 MFLOP per token-forward; 38M L=100 → 63.98
    assert abs(R.f_eff_per_sequence(R.MODEL_CONFIGS["640M"], 256) / 256 / 1e6 - 1147.5) < 0.5
    assert abs(R.f_eff_per_sequence(R.MODEL_CONFIGS["38M"], 100) / 100 / 1e6 - 63.98) < 0.05


def test_order_opt_sampler_matches_reference_trace(golden_dir):
    """generate_oaardm_order_opt (generate.py:151-177) executed by the reference's own source vs the restatement."""
    g = load(golden_dir, "order_opt_trace.npz")
    m = R.build_model(TINY_WIDE, None, 12)
    with torch.no_grad():
        m.decoder.conv.weight.mul_(float(g["decoder_scale"]))
    assert sd_digest(m.state_dict()) == str(g["sd_sha"])               # (digest of the sharpened model)
    L = len(g["uniforms"])
    trace = []
    sample = R.generate_oaardm_order_opt_ref(m, L, 2, torch.from_numpy(g["uniforms"]), trace=trace)
    assert np.array_equal(sample.numpy(), g["final"])
    assert [t["tok"] for t in trace] == g["tok"].tolist()
    assert sorted(t["pos"] for t in trace) == list(range(L))         # every position exactly once
    assert (sample[0] == sample[1]).all()                            # the batch stays identical (:174)


def test_scoring_losses_match_reference(golden_dir):
    """sum_nll_mask's model call + loss (analysis/sequence_perp.py:125-155) on the reference collaters' outputs."""
    g = load(golden_dir, "scoring.npz")
    m = R.build_model(TINY_WIDE, None, 12)
    with torch.no_grad():
        m.decoder.conv.weight.mul_(float(g["decoder_scale"]))
    for i in range(3):
        src, tgt, mask = (torch.from_numpy(g[f"oa{i}_{k}"]) for k in ("src", "tgt", "mask"))
        frac, nll, tokens = R.sum_nll_ref(m, "mask", src, torch.zeros(1, dtype=torch.long), tgt, mask)
        assert tokens == int(g[f"oa{i}_tokens"]) and abs(frac - tokens / src.shape[1]) < 1e-12
        assert abs(nll - float(g[f"oa{i}_nll"])) <= 1e-5 * abs(float(g[f"oa{i}_nll"]))
    m = R.build_model(TINY, 500, 13)
    with torch.no_grad():
        m.decoder.conv.weight.mul_(float(g["decoder_scale"]))
    for i in range(2):
        src, ts, tgt = (torch.from_numpy(g[f"d3{i}_{k}"]) for k in ("src", "timestep", "tgt"))
        t_out, nll, tokens = R.sum_nll_ref(m, "d3pm", src, ts, tgt)
        assert tokens == src.shape[1] and abs(nll - float(g[f"d3{i}_nll"])) <= 1e-5 * abs(float(g[f"d3{i}_nll"]))


def _sharp(cfg, seed, scale):
    m = R.build_model(cfg, None, seed)
    with torch.no_grad():
        m.decoder.conv.weight.mul_(float(scale))
    return m


def test_repetition_penalty_loop_matches_reference(golden_dir):
    """generate_oaardm(penalty=3) run verbatim (oracle/make_golden_r3.py): final tokens, every first draw and every
    re-draw (which step, which sequence, which token) of the oracle's restatement are the reference's."""
    g = load(golden_dir, "oadm_penalty_trace.npz")
    m = _sharp(TINY_WIDE, 12, g["decoder_scale"])
    assert sd_digest(m.state_dict()) == str(g["sd_sha"])
    B, L = g["final"].shape
    trace = []
    out = R.generate_oaardm_penalty_ref(m, L, B, g["order"], torch.from_numpy(g["uniforms"]), float(g["penalty"]), trace=trace)
    assert np.array_equal(out.numpy(), g["final"])
    redraws = [(s, j) for s, t in enumerate(trace) for j in t["redrawn"]]
    assert redraws == list(zip(g["redraw_step"].tolist(), g["redraw_seq"].tolist())) and len(redraws) >= 5
    tok = R.Tokenizer()
    assert [tok.untokenize(r) for r in out] == [str(s) for s in g["strings"]]


def test_conditional_callers_match_reference(golden_dir):
    """inpaint / generate_scaffold / generate_autoreg_scaffold run verbatim (oracle/make_golden_r3.py) against the
    oracle's loops on the same start, order and uniforms."""
    g = load(golden_dir, "conditional_trace.npz")
    m = _sharp(TINY_WIDE, 12, g["decoder_scale"])
    assert sd_digest(m.state_dict()) == str(g["sd_sha"])
    tok = R.Tokenizer()
    enc = lambda s: torch.tensor([tok.a_to_i[a] for a in s])
    for s in range(2):                      # inpaint: the region is re-sampled from the unmasked sequence around it
        seq, a, b = str(g["inpaint_seqs"][s]), int(g["inpaint_starts"][s]), int(g["inpaint_ends"][s])
        out = R.generate_oaardm_ref(m, len(seq), 1, g[f"inpaint_order{s}"], torch.from_numpy(g[f"inpaint_uni{s}"]),
                                    start=enc(seq).unsqueeze(0))
        assert tok.untokenize(out[0]) == str(g["inpaint_out_seqs"][s])
        assert tok.untokenize(out[0, a:b]) == str(g["inpaint_out_idrs"][s]) and seq[a:b] == str(g["inpaint_ref_idrs"][s])
    motif, ns, n_sc = str(g["scaffold_motif"]), int(g["scaffold_new_start"]), int(g["scaffold_length"])
    L = n_sc + len(motif)
    start = torch.full((2, L), R.MASK_ID, dtype=torch.long)
    start[:, ns:ns + len(motif)] = enc(motif)
    out = R.generate_oaardm_ref(m, L, 2, g["scaffold_order"], torch.from_numpy(g["scaffold_uni"]), start=start)
    assert [tok.untokenize(r) for r in out] == [str(s) for s in g["scaffold_strings"]]
    # causal model over the 30-token alphabet
    alphabet = str(g["ar_alphabet"])
    mc = R.build_model(dict(TINY_WIDE, n_tokens=30, causal=True), None, int(g["ar_seed"]))
    with torch.no_grad():
        mc.decoder.conv.weight.mul_(float(g["decoder_scale"]))
        mc.decoder.conv.bias[alphabet.index("*")] += float(g["ar_stop_bias"])
    assert sd_digest(mc.state_dict()) == str(g["ar_sd_sha"])
    motif, n_sc = str(g["ar_motif"]), int(g["ar_scaffold_length"])
    L = n_sc + len(motif)
    lengths = set()
    for r in range(len(g["ar_new_start"])):
        ns = int(g["ar_new_start"][r])
        ref = torch.full((L,), alphabet.index("#"), dtype=torch.long)
        ref[ns:ns + len(motif)] = torch.tensor([alphabet.index(a) for a in motif])
        trace = []
        out = R.generate_autoreg_scaffold_ref(mc, ref, alphabet.index("@"), alphabet.index("*"), alphabet.index("#"),
                                              len(alphabet), torch.from_numpy(g["ar_uni"][r]), trace=trace)
        assert "".join(alphabet[int(t)] for t in out) == str(g["ar_strings"][r])
        assert len(trace) == int(g["ar_n_draws"][r])
        assert [len(t["p"]) for t in trace] == g[f"ar{r}_n_classes"].tolist()
        lengths.add(len(out))
    assert len(lengths) > 1                 # some runs end at STOP, early
