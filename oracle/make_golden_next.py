#!/usr/bin/env python3
"""Golden fixtures for the SURVEY §8f rows 3-4 callers, produced by EXECUTING THE REFERENCE'S OWN SOURCES (test
infrastructure; runs only in the build container, see make_golden.py for the loading scheme and its limits).

* order_opt_trace.npz  - ``generate_oaardm_order_opt`` (evodiff/generate.py:151-177) run verbatim on the tiny-wide
  model with ``torch.multinomial`` replaced by the inverse CDF over pre-drawn uniforms.
* scoring.npz          - ``OAMaskCollater`` / ``D3PMCollater`` (evodiff/collaters.py) run verbatim under fixed
  seeds, the reference model's logits on their output, and the losses ``sum_nll_mask`` (analysis/sequence_perp.py:
  131-155) takes from them: ``OAMaskedCrossEntropyLoss(reweight=False)`` (evodiff/losses.py) and
  ``CrossEntropyLoss(reduction='sum')``.

Only data is written.  Usage: python oracle/make_golden_next.py [--ref /root/reference] [--out tests/golden]
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import bytenet_ref as R  # noqa: E402
from oracle.make_golden import TINY, TINY_WIDE, UniformMultinomial, _load, load_reference, ref_model, sd_digest  # noqa: E402

SEQS = ["MKVLAAGIVGLCAQEPTDSWYRNHF", "ACDEFGHIKLMNPQRSTVWYACDEFGHIKLMNPQRSTVWYMKV", "GG"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "tests", "golden"))
    args = ap.parse_args()
    torch.set_num_threads(8)
    model_mod, utils_mod, gen_mod = load_reference(args.ref)
    consts = sys.modules["sequence_models.constants"]
    consts.PAD, consts.PROTEIN_ALPHABET = "-", R.ALL_AAS + "-*#@"      # imported by collaters.py, unused on this path
    col_mod = _load("evodiff.collaters", os.path.join(args.ref, "evodiff", "collaters.py"))
    loss_mod = _load("evodiff.losses", os.path.join(args.ref, "evodiff", "losses.py"))

    # ---- order-optimised OA-DM sampler ------------------------------------------------------------
    m = ref_model(model_mod, TINY_WIDE, None, 12)
    with torch.no_grad():
        m.decoder.conv.weight.mul_(8.0)
    B, L = 2, 30
    uni = np.random.default_rng(11).random(L, dtype=np.float32)
    draw = UniformMultinomial()
    for s in range(L):
        draw.feed(uni[s:s + 1])
    real_multinomial, torch.multinomial = torch.multinomial, draw
    try:
        sample, strings = gen_mod.generate_oaardm_order_opt(m, utils_mod.Tokenizer(), L, batch_size=B, device="cpu")
    finally:
        torch.multinomial = real_multinomial
    np.savez_compressed(os.path.join(args.out, "order_opt_trace.npz"), uniforms=uni, final=sample.numpy(),
                        strings=np.array(strings), decoder_scale=8.0,
                        p=torch.stack([c[0][0] for c in draw.calls]).numpy(),
                        tok=torch.stack([c[1][0] for c in draw.calls]).numpy(),
                        margin=torch.stack([c[2][0] for c in draw.calls]).numpy(), sd_sha=sd_digest(m.state_dict()))

    # ---- scoring: collaters + losses ----------------------------------------------------------------
    out = {"seqs": np.array(SEQS), "decoder_scale": 8.0}
    tok = utils_mod.Tokenizer()
    m = ref_model(model_mod, TINY_WIDE, None, 12)
    with torch.no_grad():
        m.decoder.conv.weight.mul_(8.0)
    loss_fn = loss_mod.OAMaskedCrossEntropyLoss(reweight=False)
    for i, seq in enumerate(SEQS):
        np.random.seed(40 + i)
        src, timestep, tgt, mask = col_mod.OAMaskCollater(tok)([(seq,)])
        with torch.no_grad():
            outputs = m(src, torch.zeros(1, dtype=torch.long))
        input_mask = (src != tok.pad_id).float()
        ce, nll = loss_fn(outputs, tgt, mask, timestep, input_mask)
        out.update({f"oa{i}_src": src.numpy(), f"oa{i}_timestep": timestep.numpy(), f"oa{i}_tgt": tgt.numpy(),
                    f"oa{i}_mask": mask.numpy(), f"oa{i}_nll": float(nll), f"oa{i}_tokens": int(mask.sum())})
    # one batch of all three (padding) through the collater alone
    np.random.seed(50)
    src, timestep, tgt, mask = col_mod.OAMaskCollater(tok)([(s,) for s in SEQS])
    out.update({"oab_src": src.numpy(), "oab_timestep": timestep.numpy(), "oab_tgt": tgt.numpy(), "oab_mask": mask.numpy()})

    tok_u = utils_mod.Tokenizer(sequences=True)
    Q_bar, Q = tok_u.q_random_schedule(timesteps=500)
    m = ref_model(model_mod, TINY, 500, 13)
    with torch.no_grad():
        m.decoder.conv.weight.mul_(8.0)
    for i, seq in enumerate(SEQS[:2]):
        np.random.seed(60 + i)
        torch.manual_seed(60 + i)
        batch = col_mod.D3PMCollater(tok_u, 500, Q, Q_bar)([(seq,)])
        src, src_1h, timestep, tgt, tgt_1h, _, _, q_x = batch
        with torch.no_grad():
            outputs = m(src, timestep)
        nll = torch.nn.CrossEntropyLoss(reduction="sum")(outputs.squeeze(), tgt.squeeze()).item()
        out.update({f"d3{i}_src": src.numpy(), f"d3{i}_timestep": timestep.numpy(), f"d3{i}_tgt": tgt.numpy(),
                    f"d3{i}_qx": q_x.numpy(), f"d3{i}_nll": nll})
    np.savez_compressed(os.path.join(args.out, "scoring.npz"), **out)
    print("written: order_opt_trace.npz, scoring.npz")


if __name__ == "__main__":
    main()
