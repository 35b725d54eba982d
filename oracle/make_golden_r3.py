#!/usr/bin/env python3
"""Golden fixtures for the callers either side of the hot path, produced by EXECUTING THE REFERENCE'S OWN SOURCES (test
infrastructure; runs only in the build container, see make_golden.py for the loading scheme and its limits).

* oadm_penalty_trace.npz - ``generate_oaardm(..., penalty=p)`` (evodiff/generate.py:180-215, the repetition-penalty
  branch :202-210) run verbatim with ``torch.multinomial`` replaced by the inverse CDF over pre-drawn uniforms: the
  batched first draw of step s uses row s, the re-draw of sequence j at step s uses row n_steps + s, entry j.
* conditional_trace.npz  - ``inpaint`` (evodiff/conditional_generation.py:628-675) on two sequences of different
  length, ``generate_scaffold`` (:475-512) with a two-segment motif and batch 2, and ``generate_autoreg_scaffold``
  (:515-576) on a causal model, all run verbatim; ``get_motif`` (PDB download + cleaning) is replaced by a function
  returning the motif string, ``np.random.choice`` / ``np.random.shuffle`` run under a fixed seed.

The causal convolution behind ``generate_autoreg_scaffold`` is the oracle's restatement of the third-party
``MaskedCausalConv1d`` (parity unpinned at that boundary, oracle/bytenet_ref.py); what this fixture pins is the
reference's own loop around it.  Only data is written.
Usage: python oracle/make_golden_r3.py [--ref /root/reference] [--out tests/golden]
"""
from __future__ import annotations

import argparse
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import bytenet_ref as R  # noqa: E402
from oracle.make_golden import TINY_WIDE, _load, _module, load_reference, ref_model, sd_digest  # noqa: E402


class StepMultinomial:
    """Drop-in for torch.multinomial(p, num_samples=1) inside generate_oaardm's penalty loop: a 2-d ``p`` is the
    batched first draw of the next step; a 1-d ``p`` is the re-draw of one sequence, ``p[j]`` - a view of that step's
    (B, n) probabilities, so its storage offset tells j."""

    def __init__(self, uniforms, n_steps):
        self.u, self.n_steps, self.step, self.calls, self.redraws = torch.as_tensor(uniforms), n_steps, -1, [], []

    def __call__(self, p, num_samples=1, replacement=False):
        assert num_samples == 1
        if p.dim() == 2:
            self.step += 1
            idx, margin = R.inverse_cdf(p, self.u[self.step])
            self.calls.append((p.clone(), idx.clone(), margin.clone()))
            return idx.unsqueeze(-1)
        j = p.storage_offset() // p.numel()
        idx, margin = R.inverse_cdf(p.unsqueeze(0), self.u[self.n_steps + self.step, j].reshape(1))
        self.redraws.append((self.step, j, p.clone(), int(idx), float(margin)))
        return idx


class QueueMultinomial:
    """torch.multinomial(p, 1) -> inverse CDF over queued uniforms, in call order; keeps every call."""

    def __init__(self, queue):
        self.queue, self.calls = list(queue), []

    def __call__(self, p, num_samples=1, replacement=False):
        u = torch.as_tensor(self.queue.pop(0)).reshape(-1)
        p2 = p if p.dim() == 2 else p.unsqueeze(0)
        idx, margin = R.inverse_cdf(p2, u)
        self.calls.append((p2.clone(), idx.clone(), margin.clone()))
        return idx.unsqueeze(-1) if p.dim() == 2 else idx


@contextlib.contextmanager
def patched_multinomial(fn):
    real, torch.multinomial = torch.multinomial, fn
    try:
        with contextlib.redirect_stdout(io.StringIO()):     # the reference prints its progress
            yield
    finally:
        torch.multinomial = real


def load_conditional(ref_root):
    class _Unused:
        def __init__(self, *a, **k):
            raise RuntimeError("placeholder for an out-of-scope third-party symbol was called")
    bio = _module("biotite")
    bio.__path__ = []
    st = _module("biotite.structure", filter_peptide_backbone=_Unused, get_chains=_Unused, AtomArray=_Unused)
    st.__path__ = []
    bio.structure = st
    _module("biotite.structure.io", pdbx=_Unused, pdb=_Unused)
    _module("biotite.structure.residues", get_residues=_Unused)
    _module("biotite.sequence", ProteinSequence=_Unused)
    _module("sequence_models.utils", parse_fasta=_Unused)
    return _load("evodiff.conditional_generation", os.path.join(ref_root, "evodiff", "conditional_generation.py"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "tests", "golden"))
    args = ap.parse_args()
    torch.set_num_threads(8)
    model_mod, utils_mod, gen_mod = load_reference(args.ref)
    gen_mod.tqdm = lambda it, *a, **k: it
    cg_mod = load_conditional(args.ref)
    cg_mod.tqdm = lambda it, *a, **k: it

    # ---- repetition penalty ---------------------------------------------------------------------------------------
    m = ref_model(model_mod, TINY_WIDE, None, 12)
    with torch.no_grad():
        m.decoder.conv.weight.mul_(8.0)
    B, L, penalty = 4, 40, 3.0
    uni = np.random.default_rng(21).random((2 * L, B), dtype=np.float32)
    np.random.seed(9)
    order = np.arange(L)
    np.random.shuffle(order)
    np.random.seed(9)                                   # generate_oaardm re-draws the same permutation
    draw = StepMultinomial(uni, L)
    with patched_multinomial(draw):
        sample, strings = gen_mod.generate_oaardm(m, utils_mod.Tokenizer(), L, penalty=penalty, batch_size=B, device="cpu")
    assert len(draw.calls) == L and len(draw.redraws) >= 5, len(draw.redraws)
    np.savez_compressed(os.path.join(args.out, "oadm_penalty_trace.npz"), order=order, uniforms=uni, penalty=penalty,
                        final=sample.numpy(), strings=np.array(strings), decoder_scale=8.0,
                        first_tok=torch.stack([c[1] for c in draw.calls]).numpy(),
                        first_margin=torch.stack([c[2] for c in draw.calls]).numpy(),
                        redraw_step=np.array([r[0] for r in draw.redraws]), redraw_seq=np.array([r[1] for r in draw.redraws]),
                        redraw_tok=np.array([r[3] for r in draw.redraws]), redraw_margin=np.array([r[4] for r in draw.redraws]),
                        sd_sha=sd_digest(m.state_dict()))
    print("penalty: %d re-draws in %d steps x %d sequences" % (len(draw.redraws), L, B))

    # ---- conditional callers --------------------------------------------------------------------------------------
    out = {"decoder_scale": 8.0, "sd_sha": sd_digest(m.state_dict())}
    tok = utils_mod.Tokenizer()
    rng = np.random.default_rng(31)
    # inpaint: two sequences of different length, region re-sampled from the UNMASKED tokens around it (:641-659)
    seqs = ["".join(rng.choice(list(R.CAN_AAS), n)) for n in (57, 90)]
    starts, ends = [10, 41], [31, 77]
    uni_in = [rng.random((ends[s] - starts[s], 1), dtype=np.float32) for s in range(2)]
    np.random.seed(17)
    orders = []
    for s in range(2):                                  # the permutations inpaint will draw, in its order
        loc = np.arange(starts[s], ends[s])
        np.random.shuffle(loc)
        orders.append(loc)
    np.random.seed(17)
    draw = QueueMultinomial([u for s in range(2) for u in uni_in[s]])
    tokenized = [torch.tensor(tok.tokenizeMSA(x)) for x in seqs]
    with patched_multinomial(draw):
        res = cg_mod.inpaint(m, tokenized, starts, ends, seqs, tokenizer=tok, device="cpu")
    out.update({"inpaint_seqs": np.array(seqs), "inpaint_starts": np.array(starts), "inpaint_ends": np.array(ends),
                "inpaint_order0": orders[0], "inpaint_order1": orders[1], "inpaint_uni0": uni_in[0], "inpaint_uni1": uni_in[1],
                "inpaint_out_seqs": np.array([x[0] for x in res[0]]), "inpaint_out_idrs": np.array(res[2]),
                "inpaint_ref_idrs": np.array(res[3]),
                "inpaint_margin": torch.cat([c[2] for c in draw.calls]).numpy()})

    # generate_scaffold: two-segment motif ('#' spacer), batch 2 (every masked position visited twice, :493-495)
    motif, scaffold_length, Bs = "HELLK##AGH", 30, 2
    cg_mod.get_motif = lambda *a, **k: motif            # stands in for the PDB download + cleaning only
    Ls = scaffold_length + len(motif)
    n_mask = Bs * (Ls - 8)
    uni_sc = rng.random((n_mask, Bs), dtype=np.float32)
    np.random.seed(23)
    new_start = int(np.random.choice(scaffold_length))
    start_tokens = torch.full((Bs, Ls), tok.mask_id, dtype=torch.long)
    start_tokens[:, new_start:new_start + len(motif)] = torch.tensor(tok.tokenize((motif,)))
    loc = np.array((start_tokens == tok.mask_id).long().nonzero(as_tuple=True)[1])
    np.random.shuffle(loc)
    assert len(loc) == n_mask
    np.random.seed(23)
    draw = QueueMultinomial(list(uni_sc))
    with patched_multinomial(draw):
        strings, s_idx, e_idx = cg_mod.generate_scaffold(m, "XXXX", [0, 7], [4, 9], scaffold_length, "/nowhere/", tok,
                                                         batch_size=Bs, device="cpu")
    out.update({"scaffold_motif": motif, "scaffold_length": scaffold_length, "scaffold_new_start": new_start,
                "scaffold_order": loc, "scaffold_uni": uni_sc, "scaffold_strings": np.array(strings),
                "scaffold_start_idxs": np.array(s_idx), "scaffold_end_idxs": np.array(e_idx),
                "scaffold_margin": torch.stack([c[2] for c in draw.calls]).numpy()})

    # generate_autoreg_scaffold: causal model over the 30-token PROTEIN_ALPHABET (pretrained.py:164-173)
    cfg = dict(TINY_WIDE, n_tokens=30, causal=True)
    torch.manual_seed(33)
    tok_ar = utils_mod.Tokenizer(protein_alphabet=R.ALL_AAS + "*-#@", all_aas=R.ALL_AAS, pad="-")
    mc = model_mod.ByteNetLMTime(30, cfg["d_embed"], cfg["d_model"], cfg["n_layers"], cfg["kernel_size"], cfg["r"],
                                 causal=True, padding_idx=tok_ar.mask_id, rank=None, dropout=0.0, tie_weights=False,
                                 final_ln=True, slim=False, activation="gelu", timesteps=None).eval()
    with torch.no_grad():
        mc.decoder.conv.weight.mul_(8.0)
        mc.decoder.conv.bias[tok_ar.stop_id] += 6.0       # STOP likely enough to end some of the sequences early
    motif_ar, scaffold_ar = "WKL#PEN", 24
    cg_mod.get_motif = lambda *a, **k: motif_ar
    La = scaffold_ar + len(motif_ar)
    runs = []
    for r, seed in enumerate((41, 42, 43)):
        uni_ar = rng.random(La, dtype=np.float32)
        np.random.seed(seed)
        ns = int(np.random.choice(scaffold_ar))
        np.random.seed(seed)
        given = np.zeros(La, bool)
        given[ns:ns + len(motif_ar)] = np.array(list(motif_ar)) != "#"
        draw = QueueMultinomial([uni_ar[i:i + 1] for i in range(La) if not given[i]])
        with patched_multinomial(draw):
            strings, s_idx, e_idx = cg_mod.generate_autoreg_scaffold(mc, "XXXX", [0, 4], [2, 6], scaffold_ar, "/nowhere/",
                                                                     tok_ar, batch_size=1, device="cpu")
        runs.append((ns, uni_ar, strings[0], s_idx, e_idx, len(draw.calls),
                     np.array([float(c[2]) for c in draw.calls]), np.array([c[0].shape[-1] for c in draw.calls])))
        print("autoreg scaffold run %d: new_start %d, %d draws, length %d: %s" % (r, ns, len(draw.calls), len(strings[0]), strings[0]))
    out.update({"ar_motif": motif_ar, "ar_scaffold_length": scaffold_ar, "ar_alphabet": "".join(tok_ar.alphabet),
                "ar_sd_sha": sd_digest(mc.state_dict()), "ar_seed": 33, "ar_stop_bias": 6.0,
                "ar_new_start": np.array([r[0] for r in runs]), "ar_uni": np.stack([r[1] for r in runs]),
                "ar_strings": np.array([r[2] for r in runs]), "ar_n_draws": np.array([r[5] for r in runs])})
    for r, run in enumerate(runs):
        out.update({f"ar{r}_start_idxs": np.array(run[3]), f"ar{r}_end_idxs": np.array(run[4]), f"ar{r}_margin": run[6],
                    f"ar{r}_n_classes": run[7]})
    np.savez_compressed(os.path.join(args.out, "conditional_trace.npz"), **out)
    print("written: oadm_penalty_trace.npz, conditional_trace.npz")


if __name__ == "__main__":
    main()
