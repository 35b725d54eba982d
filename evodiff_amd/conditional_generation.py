"""Conditional OA-DM callers of the hot path.

Mirrors the sampling loops of ``evodiff/conditional_generation.py``: ``inpaint_simple`` (:603-625), ``inpaint``
(:628-675, without the ``random_baseline`` arm, which samples from a data file and never calls the model) and
``generate_scaffold`` (:475-512).  They are the loop of ``generate_oaardm`` with a partially filled start and an
order over the masked positions only, so each maps onto ONE ``ed_oadm_sample`` call (the step kernels read the
position from the order buffer; positions outside the order are never touched).

Randomness contract as in ``generate.py``: ``np.random.shuffle`` orders the positions, every categorical draw is
the inverse CDF at a pre-drawn uniform; both can be supplied (keyword-only ``order`` / ``uniforms``) for
reproducible runs and parity tests.  Reading the motif out of a PDB file (``get_motif``) is outside the hot path:
``generate_scaffold`` takes the motif sequence itself.
"""
import numpy as np
import torch

from .utils import Tokenizer


def _n_draw(tokenizer):
    return len(tokenizer.all_aas) - 6              # only canonical amino acids are ever predicted


def _fill(model, sample, loc, uniforms, tokenizer):
    """Unmask the positions ``loc`` of ``sample`` (B, L) in that order, in place, on the GPU path."""
    n_steps, B = len(loc), sample.shape[0]
    if n_steps == 0:
        return
    if uniforms is None:
        uniforms = torch.rand((n_steps, B), dtype=torch.float32, device=sample.device)
    uniforms = torch.as_tensor(uniforms, dtype=torch.float32).to(sample.device)
    with torch.no_grad():
        model.oadm_sample(sample, torch.as_tensor(np.asarray(loc), dtype=torch.int32), uniforms[:n_steps],
                          _n_draw(tokenizer))


def get_intervals(positions, single_res_domain=False):
    """Start and end index of every run of consecutive positions (conditional_generation.py:456-472; the end of a
    run is its last position, as there)."""
    positions = [int(p) for p in positions]
    if single_res_domain:
        return list(positions), list(positions)
    start, stop = [], []
    for i, item in enumerate(positions):
        if i == 0:
            start.append(item)
        elif i == len(positions) - 1:
            stop.append(item)
        elif item + 1 != positions[i + 1]:
            stop.append(item)
            start.append(positions[i + 1])
    return start, stop


def inpaint_simple(model, sequence, start_idx, end_idx, tokenizer=Tokenizer(), device='cuda', *, order=None,
                   uniforms=None):
    """Re-sample ``sequence[start_idx:end_idx]`` given the rest.  Returns (tokens (L,), sequence, region)."""
    idr_length = end_idx - start_idx
    masked_sequence = sequence[0:start_idx] + '#' * idr_length + sequence[end_idx:]
    sample = torch.tensor(tokenizer.tokenizeMSA(masked_sequence), dtype=torch.long, device=device).unsqueeze(0)
    sample = sample.contiguous()
    if order is None:
        loc = np.arange(start_idx, end_idx)
        np.random.shuffle(loc)
    else:
        loc = np.asarray(order)
    _fill(model, sample, loc, uniforms, tokenizer)
    sample = sample[0]
    return sample, tokenizer.untokenize(sample), tokenizer.untokenize(sample[start_idx:end_idx])


def inpaint(model, tokenized_sequences, start_idxs, end_idxs, sequences, tokenizer=Tokenizer(), device='cuda',
            random_baseline=False, data_top_dir='/', *, orders=None, uniforms=None):
    """``inpaint_simple`` over a list of tokenised sequences (one call per sequence: the lengths differ)."""
    if random_baseline:
        raise NotImplementedError("random_baseline samples from a data file and never calls the model")
    samples, samples_idr = [], []
    for s, tokens in enumerate(tokenized_sequences):
        sample = torch.as_tensor(tokens).to(device=device, dtype=torch.long).clone().unsqueeze(0).contiguous()
        if orders is None:
            loc = np.arange(start_idxs[s], end_idxs[s])
            np.random.shuffle(loc)
        else:
            loc = np.asarray(orders[s])
        _fill(model, sample, loc, None if uniforms is None else uniforms[s], tokenizer)
        samples.append(sample[0])
        samples_idr.append(sample[0, start_idxs[s]:end_idxs[s]])
    untokenized_seqs = [[tokenizer.untokenize(s)] for s in samples]
    untokenized_idrs = [tokenizer.untokenize(s) for s in samples_idr]
    sequences_idrs = [sequences[s][start_idxs[s]:end_idxs[s]] for s in range(len(samples))]
    return (untokenized_seqs, [[s] for s in sequences], untokenized_idrs, sequences_idrs, list(start_idxs),
            list(end_idxs))


def generate_scaffold(model, motif_seq, scaffold_length, tokenizer, batch_size=1, device='cuda',
                      single_res_domain=False, *, new_start=None, order=None, uniforms=None):
    """Place ``motif_seq`` (may contain '#' spacers, as ``get_motif`` returns it) at a random offset of an
    all-mask scaffold and fill every masked position.  Returns (sequences, motif start idxs, motif end idxs)."""
    mask = tokenizer.mask_id
    motif_tokenized = torch.as_tensor(tokenizer.tokenize((motif_seq,)), dtype=torch.long)
    seq_len = scaffold_length + len(motif_seq)
    sample = torch.full((batch_size, seq_len), mask, dtype=torch.long)
    if new_start is None:
        new_start = np.random.choice(scaffold_length)      # randomly place the motif in the scaffold
    sample[:, new_start:new_start + len(motif_seq)] = motif_tokenized
    nonmask_locations = (sample[0] != mask).nonzero().flatten()
    new_start_idxs, new_end_idxs = get_intervals(nonmask_locations, single_res_domain=single_res_domain)
    if order is None:
        # as the reference (:493-495): the column indices of the masked entries of the WHOLE batch, so with
        # batch_size > 1 every masked position is visited (and re-drawn) batch_size times
        loc = (sample == mask).long().nonzero(as_tuple=True)[1].numpy()
        np.random.shuffle(loc)
    else:
        loc = np.asarray(order)
    sample = sample.to(device).contiguous()
    _fill(model, sample, loc, uniforms, tokenizer)
    untokenized = [tokenizer.untokenize(s) for s in sample]
    return untokenized, new_start_idxs, new_end_idxs
