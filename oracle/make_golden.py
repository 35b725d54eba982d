#!/usr/bin/env python3
"""Generate tests/golden/*.npz by EXECUTING THE REFERENCE'S OWN SOURCES (test infrastructure).

Runs only in the build container (needs /root/reference; the GPU box never has it).  The
reference package cannot be imported as a package (``sequence_models``, ``esm``, ``Bio`` are not
installed and cannot be installed offline), so ``evodiff/model.py``, ``evodiff/utils.py`` and
``evodiff/generate.py`` are loaded *by file path* after registering minimal modules for their
missing third-party imports:

* ``sequence_models.{layers,convolutional,constants}`` → the restatements in
  ``oracle/bytenet_ref.py`` (that boundary is "parity unpinned", see its header);
* ``esm.modules``, ``Bio``, ``sequence_models.datasets``, ``evodiff.plot``, ``evodiff.pretrained``
  → empty placeholders (never called on this path).

What the fixtures therefore pin is everything the reference itself owns: PositionalEncoding1D,
the embed/time-encoding reshape quirk, the dilation schedule, block wiring, final LayerNorm and
decoder, the state_dict key layout, Tokenizer / BLOSUM / uniform schedules, and the
``generate_oaardm`` / ``generate_d3pm`` loops executed verbatim with ``torch.multinomial``
replaced by an inverse-CDF draw over pre-drawn uniforms (north_star's sampling contract).

Only data (inputs and outputs) is written; no reference source text is copied.

Usage:  python oracle/make_golden.py [--ref /root/reference] [--out tests/golden]
"""
from __future__ import annotations

import argparse
import hashlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import bytenet_ref as R  # noqa: E402


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference(ref_root):
    """Returns (model_mod, utils_mod, generate_mod) = the reference's own modules."""
    class _Unused:  # placeholder for symbols never touched on the sequence hot path
        def __init__(self, *a, **k):
            raise RuntimeError("placeholder for an out-of-scope third-party symbol was called")

    consts = dict(MSA_PAD=R.MSA_PAD, MASK=R.MASK, MSA_ALPHABET=R.MSA_ALPHABET, MSA_AAS=R.MSA_AAS,
                  GAP=R.GAP, START=R.START, STOP=R.STOP, SEP=R.SEP, AAINDEX_ALPHABET=R.AAINDEX_ALPHABET,
                  AMB_AAS=R.AMB_AAS, OTHER_AAS=R.OTHER_AAS)
    _module("sequence_models").__path__ = []
    _module("sequence_models.constants", **consts)
    _module("sequence_models.layers", PositionFeedForward=R.PositionFeedForward, DoubleEmbedding=_Unused)
    _module("sequence_models.convolutional", ByteNetBlock=R.ByteNetBlock)
    _module("sequence_models.datasets", UniRefDataset=_Unused)
    _module("esm").__path__ = []
    _module("esm.modules", TransformerLayer=_Unused, LearnedPositionalEmbedding=_Unused,
            RobertaLMHead=_Unused, ESM1bLayerNorm=_Unused, AxialTransformerLayer=_Unused)
    _module("Bio").__path__ = []
    _module("Bio.PDB", PDBParser=_Unused)
    pkg = _module("evodiff")
    pkg.__path__ = []
    _module("evodiff.plot", aa_reconstruction_parity_plot=_Unused)
    pre = _module("evodiff.pretrained")
    pre.__getattr__ = lambda name: _Unused          # generate.py:13 imports loader names only
    src = os.path.join(ref_root, "evodiff")
    _load("evodiff.constants", os.path.join(src, "constants.py"))
    utils = _load("evodiff.utils", os.path.join(src, "utils.py"))
    model = _load("evodiff.model", os.path.join(src, "model.py"))
    gen = _load("evodiff.generate", os.path.join(src, "generate.py"))
    return model, utils, gen


def sd_digest(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(np.ascontiguousarray(v.detach().numpy()).tobytes())
    return h.hexdigest()


def arr_digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def ref_model(model_mod, cfg, timesteps, seed):
    """pretrained.py:39-42 constructor call on the reference's own ByteNetLMTime."""
    torch.manual_seed(seed)
    return model_mod.ByteNetLMTime(len(R.MSA_ALPHABET), cfg["d_embed"], cfg["d_model"], cfg["n_layers"],
                                   cfg["kernel_size"], cfg["r"], causal=False, padding_idx=R.MASK_ID,
                                   rank=None, dropout=0.0, tie_weights=False, final_ln=True,
                                   slim=cfg["slim"], activation=cfg["activation"], timesteps=timesteps).eval()


class UniformMultinomial:
    """Drop-in for torch.multinomial(p, num_samples=1): inverse-CDF over queued uniforms."""

    def __init__(self):
        self.queue, self.calls = [], []

    def feed(self, u):
        self.queue.append(torch.as_tensor(u))

    def __call__(self, p, num_samples=1, replacement=False):
        assert num_samples == 1
        u = self.queue.pop(0)
        squeeze = p.dim() == 1
        p2 = p.unsqueeze(0) if squeeze else p
        idx, margin = R.inverse_cdf(p2, u.reshape(p2.shape[0]))
        self.calls.append((p2.clone(), idx.clone(), margin.clone()))
        idx = idx.unsqueeze(-1)
        return idx.squeeze(0) if squeeze else idx


TINY = dict(d_embed=8, d_model=64, n_layers=8, kernel_size=5, r=128, slim=True, activation="gelu")
TINY_WIDE = dict(d_embed=8, d_model=256, n_layers=9, kernel_size=5, r=128, slim=False, activation="gelu")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "tests", "golden"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    torch.set_num_threads(8)
    model_mod, utils_mod, gen_mod = load_reference(args.ref)
    gen_mod.tqdm = lambda it, *a, **k: it                      # silence progress bars only

    # ---- (i) tiny models: weights, inputs, per-block outputs, logits --------------------------
    for tag, cfg, seed in (("tiny", TINY, 11), ("tinywide", TINY_WIDE, 12)):
        m = ref_model(model_mod, cfg, None, seed)
        sd = m.state_dict()
        mine = R.build_model(cfg, None, seed)
        assert list(mine.state_dict().keys()) == list(sd.keys())
        assert sd_digest(mine.state_dict()) == sd_digest(sd), "seeded init differs from reference ctor"
        out = {"sd_sha": sd_digest(sd), "keys": np.array(list(sd.keys()))}
        if tag == "tiny":                  # small enough to keep verbatim; others regenerate from seed
            out.update({"sd/" + k: v.numpy() for k, v in sd.items()})
        rng = np.random.default_rng(100)
        for L in (7, 100, 300):
            x = torch.from_numpy(rng.integers(0, 31, (2, L)))
            x[0, : L // 2] = R.MASK_ID
            with torch.no_grad():
                y = m(x, torch.zeros(2, dtype=torch.long))
                e = m.embedder._embed(x, None, timesteps=None)
                blocks = [e]
                for layer in m.embedder.layers:
                    e = layer(e)
                    blocks.append(e)
            out[f"x_L{L}"] = x.numpy()
            out[f"logits_L{L}"] = y.numpy()
            out[f"blocks_L{L}"] = torch.stack(blocks).numpy() if (L <= 100 and tag == "tiny") else np.zeros(0, np.float32)
        np.savez_compressed(os.path.join(args.out, f"{tag}_forward.npz"), **out)

    # ---- (vi) time-encoding scramble with non-uniform y (quirk Q1) -----------------------------
    m = ref_model(model_mod, TINY, 500, 13)
    x = torch.from_numpy(np.random.default_rng(101).integers(0, 26, (3, 5)))
    y = torch.tensor([499, 7, 250])
    with torch.no_grad():
        np.savez_compressed(os.path.join(args.out, "tiny_timeenc.npz"), x=x.numpy(), y=y.numpy(),
                            embed=m.embedder._embed(x, y, timesteps=500).numpy(), logits=m(x, y).numpy(),
                            pe=model_mod.PositionalEncoding1D(8, 500)(torch.arange(500)).numpy(),
                            sd_sha=sd_digest(m.state_dict()))

    # ---- (ii) 38M seeded: logits for B=2, L=100 -----------------------------------------------
    cfg = R.MODEL_CONFIGS["38M"]
    m = ref_model(model_mod, cfg, None, 0)
    n_params = sum(p.numel() for p in m.parameters())
    x = torch.from_numpy(np.random.default_rng(102).integers(0, 27, (2, 100)))
    x[1, 30:80] = R.MASK_ID
    with torch.no_grad():
        y = m(x, torch.zeros(2, dtype=torch.long))
    np.savez_compressed(os.path.join(args.out, "m38_forward.npz"), x=x.numpy(), logits=y.numpy(),
                        n_params=n_params, sd_sha=sd_digest(m.state_dict()), n_tensors=len(m.state_dict()))
    print("38M params", n_params, "tensors", len(m.state_dict()))

    # ---- (iii) transition schedules -------------------------------------------------------------
    blosum = os.path.join(args.ref, "data", "blosum62-special-MSA.mat")
    tok_b = utils_mod.Tokenizer(path_to_blosum=blosum, sequences=True)
    Qb_bar, Qb = tok_b.q_blosum_schedule(timesteps=500)
    tok_u = utils_mod.Tokenizer(sequences=True)
    Qu_bar, Qu = tok_u.q_random_schedule(timesteps=500)
    rows = np.array([0, 1, 2, 250, 497, 498, 499])
    np.savez_compressed(os.path.join(args.out, "schedules.npz"), rows=rows,
                        blosum_q=tok_b.q_blosum().numpy(),
                        blosum_Q=Qb.numpy()[rows], blosum_Qbar=Qb_bar.numpy()[rows],
                        uniform_Q=Qu.numpy()[rows], uniform_Qbar=Qu_bar.numpy()[rows],
                        blosum_Q_sha=arr_digest(Qb.numpy()), blosum_Qbar_sha=arr_digest(Qb_bar.numpy()),
                        uniform_Q_sha=arr_digest(Qu.numpy()), uniform_Qbar_sha=arr_digest(Qu_bar.numpy()),
                        K=tok_b.K, mask_id=utils_mod.Tokenizer().mask_id, K_msa=utils_mod.Tokenizer().K,
                        alphabet="".join(utils_mod.Tokenizer().alphabet))

    # ---- (iv) OA-DM trace: reference generate_oaardm verbatim, B=2, L=50, tiny-wide model ------
    m = ref_model(model_mod, TINY_WIDE, None, 12)
    with torch.no_grad():                 # make the softmax non-trivial: sharpen the decoder
        m.decoder.conv.weight.mul_(8.0)
    B, L = 2, 50
    uni = np.random.default_rng(1).random((L, B), dtype=np.float32)
    draw = UniformMultinomial()
    for s in range(L):
        draw.feed(uni[s])
    np.random.seed(5)
    order = np.arange(L)
    np.random.shuffle(order)
    np.random.seed(5)                      # generate_oaardm re-draws the same permutation
    real_multinomial, torch.multinomial = torch.multinomial, draw
    try:
        sample, strings = gen_mod.generate_oaardm(m, utils_mod.Tokenizer(), L, batch_size=B, device="cpu")
    finally:
        torch.multinomial = real_multinomial
    np.savez_compressed(os.path.join(args.out, "oadm_trace.npz"), order=order, uniforms=uni,
                        final=sample.numpy(), strings=np.array(strings), decoder_scale=8.0,
                        p=torch.stack([c[0] for c in draw.calls]).numpy(),
                        tok=torch.stack([c[1] for c in draw.calls]).numpy(),
                        margin=torch.stack([c[2] for c in draw.calls]).numpy(),
                        sd_sha=sd_digest(m.state_dict()))

    # ---- (v) D3PM trace: reference generate_d3pm verbatim, 499 steps, B=2, L=20 ----------------
    for tag, tok, Q, Qbar in (("blosum", tok_b, Qb, Qb_bar), ("uniform", tok_u, Qu, Qu_bar)):
        m = ref_model(model_mod, TINY, 500, 13)
        with torch.no_grad():
            m.decoder.conv.weight.mul_(8.0)
        B, L, T = 2, 20, 500
        uni = np.random.default_rng(2).random((T, B, L), dtype=np.float32)
        start = np.random.default_rng(3).integers(0, tok.K, (B, L))
        draw = UniformMultinomial()
        for s in range(T - 1):
            for b in range(B):
                draw.feed(uni[s, b])
                if s == T - 2:             # t == 1: the second draw over K-6 categories
                    draw.feed(uni[T - 1, b])
        real_multinomial, real_randint = torch.multinomial, torch.randint
        torch.multinomial = draw
        torch.randint = lambda *a, **k: torch.from_numpy(start.copy())
        try:
            sample, strings = gen_mod.generate_d3pm(m, tok, Q, Qbar, T, L, batch_size=B, device="cpu")
        finally:
            torch.multinomial, torch.randint = real_multinomial, real_randint
        calls = draw.calls
        assert len(calls) == (T - 1) * B + B
        keep = [0, 1, 2, 250, 497]          # step indices s (t = 499 - s) stored in full
        post = np.stack([np.stack([calls[s * B + b][0].numpy() for b in range(B)]) for s in keep])
        toks = np.stack([np.stack([calls[s * B + b][1].numpy() for b in range(B)]) for s in range(T - 2)])
        last = calls[(T - 2) * B:]           # t == 1: (full, restricted) per batch item
        np.savez_compressed(os.path.join(args.out, f"d3pm_trace_{tag}.npz"), uniforms=uni, start=start,
                            final=sample.numpy(), strings=np.array(strings), keep=np.array(keep),
                            post=post, toks=toks, decoder_scale=8.0,
                            last_post=np.stack([last[2 * b + 1][0].numpy() for b in range(B)]),
                            sd_sha=sd_digest(m.state_dict()))
    print("golden fixtures written to", args.out)


if __name__ == "__main__":
    main()
