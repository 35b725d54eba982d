"""Perplexity scoring of one sequence: ``sum_nll_mask`` of ``analysis/sequence_perp.py:101-156``
for the ByteNet schemes ('mask' = OA-DM, 'd3pm').  The collater corrupts the sequence on the host as in the
reference; forward + cross entropy run as ONE call into libevodiff_hip.so (``ed_score_nll``: the per-position NLL
comes back, the logits never leave the GPU).  The 'causal-mask' (LR-AR) and 'esm-mask' schemes are other models.
"""
import numpy as np
import torch


def sum_nll_mask(sequence, checkpoint, device='cuda', *, collated=None):
    """``sequence``: a list holding one sequence string; ``checkpoint``: a loader's tuple.  Returns
    (timestep sampled or fraction masked, summed NLL, tokens scored) like the reference.  ``collated`` (keyword-only,
    not in the reference) injects a collater output for parity tests."""
    model, collater, tokenizer, scheme = checkpoint[:4]
    model.eval()
    batch = collater(sequence) if collated is None else collated
    if scheme == 'd3pm':
        src, _, timestep, tgt = batch[:4]
        src, timestep, tgt = src.to(device), timestep.to(device), tgt.to(device)
        with torch.no_grad():
            nll = model.score_nll(src, timestep, tgt)              # CrossEntropyLoss(reduction='sum'), every logit
        return timestep, float(nll.sum(dtype=torch.float32)), int(tgt.squeeze(0).numel())
    if scheme == 'mask':
        src, _, tgt, mask = batch
        timestep = torch.zeros(len(src), dtype=torch.long, device=device)   # placeholder in the model
        src, tgt, mask = src.to(device), tgt.to(device), mask.to(device)
        with torch.no_grad():
            nll = model.score_nll(src, timestep, tgt, select=mask)  # OAMaskedCrossEntropyLoss(reweight=False)
        tokens = int(mask.sum().item())
        return tokens / int(tgt.squeeze(0).numel()), float(nll.sum(dtype=torch.float32).to(torch.float64)), tokens
    raise NotImplementedError(f"scheme {scheme!r}: only the ByteNet 'mask' and 'd3pm' checkpoints are scored here")


def perplexity(losses, n_tokens):
    """exp(sum of NLLs / sum of tokens) (sequence_perp.py:93)."""
    return float(np.exp(sum(losses) / sum(n_tokens)))
